"""Pins the oracle's Philox4x32-10 against the Random123 known-answer vectors."""
import numpy as np

from oracle import vmc_oracle as vo


def _kat(ctr, key):
  return [int(v) for v in vo.philox4x32_10(*[np.uint32(c) for c in ctr],
                                           *[np.uint32(k) for k in key])]


def test_philox_known_answers():
  # Random123 kat_vectors: philox4x32 10
  assert _kat((0, 0, 0, 0), (0, 0)) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
  assert _kat((0xffffffff,) * 4, (0xffffffff,) * 2) == [
      0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
  assert _kat((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344),
              (0xa4093822, 0x299f31d0)) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_uniform_range_and_shapes():
  u, ua = vo.step_uniforms(2024, np.arange(64), 7, 10)
  assert u.shape == (64, 10) and ua.shape == (64,)
  assert u.dtype == np.float32 and (u >= 0).all() and (u < 1).all()
  # different chains / steps give different streams
  u2, _ = vo.step_uniforms(2024, np.arange(64), 8, 10)
  assert not np.array_equal(u, u2)
  # chain-id keyed: a shard sees the same numbers as the full batch
  u3, ua3 = vo.step_uniforms(2024, np.arange(32, 64), 7, 10)
  assert np.array_equal(u3, u[32:]) and np.array_equal(ua3, ua[32:])


def test_proposal_picks_one_up_one_down():
  rng = np.random.RandomState(0)
  cfg = vo.random_configurations(12, 50, rng)
  u, _ = vo.step_uniforms(1, np.arange(50), 0, 12)
  i_up, i_dn = vo.propose_exchange(cfg, u)
  rows = np.arange(50)
  assert (cfg[rows, i_up] == 1).all() and (cfg[rows, i_dn] == -1).all()


def test_tie_event_fixture_matches_the_philox_stream():
  """tests/golden/tie_events.json (gen_tie_events.py): the listed chains really draw their largest
  site uniform twice at the listed step, and np.argmax / np.argmin break the tie by first index."""
  import json
  import os
  here = os.path.dirname(os.path.abspath(__file__))
  with open(os.path.join(here, 'golden', 'tie_events.json')) as f:
    ties = json.load(f)
  assert len(ties['events']) >= 4
  for e in ties['events']:
    u, _ = vo.step_uniforms(ties['seed'], np.array([e['chain']], np.uint32), e['step'], e['n_sites'])
    top = np.nonzero(u[0] == u[0].max())[0]
    assert [int(s) for s in top] == e['sites'] and len(top) >= 2
    up = np.ones((1, e['n_sites']), np.float32)
    assert vo.propose_exchange(up, u)[0][0] == e['sites'][0]
    assert vo.propose_exchange(-up, u)[1][0] == e['sites'][0]


def test_sortable_keys_order_like_argmax_with_first_index_ties():
  """The production sampler reduces keys (24-bit draw << 8) | (255 - site) with integer max
  (csrc/sweep16.hpp: to_keys / keyed_publish) instead of argmax / argmin of s*u.  Restated in numpy
  on draws with many forced ties: both give the same (i_up, i_dn) as propose_exchange."""
  rng = np.random.default_rng(7)
  for n in (16, 100, 256):
    draws = rng.integers(0, 1 << 24, size=(2000, n), dtype=np.uint32)
    draws[:, : n // 2] >>= rng.integers(14, 24, size=(2000, 1)).astype(np.uint32)  # few distinct values: ties
    rng.permuted(draws, axis=1, out=draws)
    u = draws.astype(np.float32) * np.float32(1.0 / 16777216.0)
    spins = np.where(rng.permuted(np.tile(np.arange(n) % 2, (2000, 1)), axis=1) == 0, 1.0, -1.0).astype(np.float32)
    i_up, i_dn = vo.propose_exchange(spins, u)
    keys = (draws.astype(np.uint64) << np.uint64(8)) | (np.uint64(255) - np.arange(n, dtype=np.uint64))[None, :]
    kup = np.where(spins > 0, keys, 0).max(axis=1)
    kdn = np.where(spins < 0, keys, 0).max(axis=1)
    np.testing.assert_array_equal(255 - (kup & np.uint64(255)).astype(np.int64), i_up)
    np.testing.assert_array_equal(255 - (kdn & np.uint64(255)).astype(np.int64), i_dn)
