"""MI355X-native batched VMC inner loop of cgs-vmc behind the reference's Python API.

Modules mirror /root/reference/cgs_vmc: utils, layers, wavefunctions, operators,
graph_builders, training, evaluation (+ session: the tf.Session stand-in, engine: the
C-ABI wrapper, parallel: chain sharding over RCCL).
"""
__all__ = ['utils', 'layers', 'wavefunctions', 'operators', 'graph_builders', 'training',
           'evaluation', 'session', 'engine', 'parallel', 'lattice']
