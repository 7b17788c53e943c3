"""Philox4x32-10 known-answer tests (Random123 kat_vectors) + noise helpers."""
import numpy as np

from oracle import sisua_oracle as so


def _kat(ctr, key):
  return [int(v[0]) for v in so.philox4x32_10(*[np.array([c]) for c in ctr], *key)]


def test_philox_known_answers():
  assert _kat((0, 0, 0, 0), (0, 0)) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
  f = 0xffffffff
  assert _kat((f, f, f, f), (f, f)) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
  assert _kat((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == \
      [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_dropout_mask_statistics_and_determinism():
  ids = np.arange(512)
  m1 = so.philox_dropout_mask(8, so.STREAM_ENC_DROPOUT, 3, ids, 130, 0.1)
  m2 = so.philox_dropout_mask(8, so.STREAM_ENC_DROPOUT, 3, ids, 130, 0.1)
  assert np.array_equal(m1, m2)
  assert m1.shape == (512, 130)
  keep = (m1 > 0).mean()
  assert abs(keep - 0.9) < 0.01
  assert np.allclose(m1[m1 > 0], 1.0 / 0.9, rtol=1e-6)
  # keyed by cell id, not by position in the batch -> sharding-independent
  perm = np.random.RandomState(0).permutation(512)
  m3 = so.philox_dropout_mask(8, so.STREAM_ENC_DROPOUT, 3, ids[perm], 130, 0.1)
  assert np.array_equal(m3, m1[perm])


def test_normal_moments():
  n = so.philox_normal(8, so.STREAM_EPS_Z, 0, np.arange(4096), 30)
  assert n.shape == (4096, 30)
  assert abs(n.mean()) < 0.01 and abs(n.std() - 1.0) < 0.01
  assert abs((n ** 3).mean()) < 0.05 and abs((n ** 4).mean() - 3.0) < 0.1
