"""NativeRandomState: the subset of numpy.random.RandomState the loop's sampler uses (uniform, choice(replace=False)),
served by libnpp_hip.so's host-side MT19937 (csrc/npp_host_rng.hip) -- the same stream as NumPy for the same seed, several
times faster, and GIL-free (a ctypes call), so a producer thread can run it beside the training loop."""
import ctypes as C

import numpy as np

from ._lib import lib, check


class NativeRandomState:
    def __init__(self, seed=0):
        self._L = lib()
        self._h = C.c_void_p(self._L.npp_rng_create(int(seed) & 0xffffffff))
        if not self._h:
            raise MemoryError("npp_rng_create")
        self._scratch = np.empty(0, np.int64)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.npp_rng_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def seed(self, seed):
        check(self._L.npp_rng_seed(self._h, int(seed) & 0xffffffff), "npp_rng_seed")

    def get_state(self):
        key = np.empty(624, np.uint32)
        pos = C.c_int32(0)
        check(self._L.npp_rng_get_state(self._h, key.ctypes.data_as(C.c_void_p), C.byref(pos)), "npp_rng_get_state")
        return ("MT19937", key, int(pos.value), 0, 0.0)

    def set_state(self, state):
        key = np.ascontiguousarray(state[1], np.uint32)
        check(self._L.npp_rng_set_state(self._h, key.ctypes.data_as(C.c_void_p), int(state[2])), "npp_rng_set_state")

    def uniform(self, low=0.0, high=1.0, size=None):
        if size is None:
            return float(self._L.npp_rng_uniform(self._h, float(low), float(high)))
        n = int(np.prod(size))
        return np.array([self._L.npp_rng_uniform(self._h, float(low), float(high)) for _ in range(n)]).reshape(size)

    def choice(self, a, size=None, replace=True, p=None):
        if replace or p is not None or not isinstance(a, (int, np.integer)):
            raise NotImplementedError("only choice(n, size, replace=False) is on the loop's path")
        n = int(a)
        shape = (int(size),) if isinstance(size, (int, np.integer)) else tuple(int(s) for s in size)
        k = int(np.prod(shape))
        if self._scratch.shape[0] < n:
            self._scratch = np.empty(n, np.int64)
        out = np.empty(k, np.int64)
        check(self._L.npp_rng_choice_noreplace(self._h, n, k, self._scratch.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)),
              "npp_rng_choice_noreplace")
        return out.reshape(shape)
