"""CPU: session.LazySoftmax - the array-like the rec session returns in the place of the softmax ndarray - against
numpy itself, with a stand-in for the device side: the two reductions rapidocr's CTCLabelDecode performs (argmax / max over the class
axis, rapid_ocr.py:443-449) answer without materialising; every other access yields exactly the wrapped array."""
import numpy as np
import torch

from rapiddoc_amd.session import LazySoftmax


class _FakeSession:
    def __init__(self):
        self.softmax_materialized = 0
        self.copies = 0

    def _to_host(self, t, copy_out=None):
        self.copies += 1
        return t.numpy().copy()


def _lazy(a, sess=None):
    sess = sess or _FakeSession()
    t = torch.from_numpy(a)
    return LazySoftmax(sess, t, torch.from_numpy(a.argmax(2).astype(np.int32)), torch.from_numpy(a.max(2))), sess


def test_the_ctc_decodes_two_reductions_do_not_materialise():
    a = np.random.default_rng(0).random((6, 40, 97)).astype(np.float32)
    l, s = _lazy(a)
    for ax in (2, -1):
        assert np.array_equal(l.argmax(axis=ax), a.argmax(axis=ax)) and l.argmax(axis=ax).dtype == a.argmax(axis=ax).dtype
        assert np.array_equal(l.max(axis=ax), a.max(axis=ax)) and l.max(axis=ax).dtype == a.dtype
        assert np.array_equal(np.argmax(l, axis=ax), a.argmax(axis=ax)) and np.array_equal(np.max(l, axis=ax), a.max(axis=ax))
    assert not l.materialized and s.copies == 0 and s.softmax_materialized == 0
    assert l.shape == a.shape and l.ndim == 3 and l.dtype == a.dtype and len(l) == 6 and l.size == a.size and l.nbytes == a.nbytes
    assert not l.materialized
    r = l.argmax(axis=2)
    r[:] = -1                                             # the caller's array, not the object's state
    assert np.array_equal(l.argmax(axis=2), a.argmax(axis=2))


def test_every_other_access_is_the_exact_array():
    a = np.random.default_rng(1).random((3, 5, 11)).astype(np.float32)
    cases = [lambda x: np.asarray(x), lambda x: np.array(x), lambda x: x[1], lambda x: x[:, 2:4, ::3], lambda x: x + 1, lambda x: 2.0 * x,
             lambda x: x.sum(axis=2), lambda x: np.sum(x), lambda x: x.argmax(axis=1), lambda x: x.max(), lambda x: x.argmax(),
             lambda x: np.exp(x), lambda x: x.reshape(15, 11), lambda x: x.T, lambda x: np.concatenate([x, x], axis=0),
             lambda x: x.astype(np.float64), lambda x: np.ascontiguousarray(x), lambda x: list(x)[2], lambda x: x.tobytes(),
             lambda x: np.argmax(x, axis=0), lambda x: x > 0.5, lambda x: np.where(x > 0.5, x, 0), lambda x: x.mean(axis=-1)]
    for f in cases:
        l, s = _lazy(a)
        got, want = f(l), f(a)
        assert l.materialized and s.copies == 1 and s.softmax_materialized == 1
        if isinstance(want, bytes):
            assert got == want
        else:
            assert type(got) is type(want) and np.array_equal(got, want)
            assert getattr(got, "dtype", None) == getattr(want, "dtype", None)
    l, s = _lazy(a)
    np.asarray(l); l[0]; l + 1
    assert s.copies == 1                                  # copied off the device once
    l[0, 0, 0] = 5.0
    assert np.asarray(l)[0, 0, 0] == 5.0
    assert l.argmax(axis=2)[0, 0] == 0                    # once materialised, reductions follow the array (here: the write)


def test_ties_follow_numpys_rule_when_the_device_reports_the_first_index():
    a = np.zeros((1, 2, 5), np.float32)
    a[0, 0, [1, 3]] = 0.5
    a[0, 1, [4]] = 0.25
    l, _ = _lazy(a)
    assert l.argmax(axis=2).tolist() == [[1, 4]] == a.argmax(axis=2).tolist()
