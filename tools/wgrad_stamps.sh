#!/bin/bash
# GPU box helper: the weight-gradient kernel's in-kernel timeline (diagnostic build -DVMC_WGRAD_STAMPS:
# wall_clock64 stamps of every workgroup at loop start / loop end / stores drained / ticket / fold end).
# Compiles a stamped grad.o next to the product objects, links a stamped library in place of the product
# one, runs a short bench, restores the product library.  Never quote run times of the stamped build.
set -uo pipefail
cd "$(dirname "$0")/.."
C=cgs_vmc_amd/csrc
cp cgs_vmc_amd/libcgsvmc_hip.so /tmp/lib_product.so
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DVMC_WGRAD_STAMPS -c $C/grad.hip -o /tmp/grad_stamped.o || exit 1
OBJS=$(ls $C/*.o | grep -v "/grad.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS /tmp/grad_stamped.o -o cgs_vmc_amd/libcgsvmc_hip.so || exit 1
python bench.py --steps 40 --warmup 5 --reps 1 --no-cpu-baseline --no-timing "$@" 2>&1 | grep -v "^{" | tail -6 | cut -c1-400
cp /tmp/lib_product.so cgs_vmc_amd/libcgsvmc_hip.so
