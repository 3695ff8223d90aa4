"""ctypes wrapper over oracle/replay_ref.c.  TEST INFRASTRUCTURE."""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class MT(ctypes.Structure):
    _fields_ = [("mt", ctypes.c_uint32 * 624), ("pos", ctypes.c_int32)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libreplay_ref.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.mt_next32.restype = ctypes.c_uint32
        L.np_next_double.restype = ctypes.c_double
        L.py_randint.restype = ctypes.c_int64
        L.py_randint.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64]
        L.np_randint.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]
        L.np_uniform.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_int64]
        L.replay_sample_seq.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
                                        ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
        L.replay_sample_mixed.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
                                          ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                          ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
        _LIB = L
    return _LIB


def np_seeded(seed: int) -> MT:
    s = MT()
    lib().mt_init_genrand(ctypes.byref(s), ctypes.c_uint32(seed))
    return s


def py_seeded(seed: int) -> MT:
    s = MT()
    seed = abs(int(seed))
    words = []
    while True:
        words.append(seed & 0xFFFFFFFF)
        seed >>= 32
        if not seed:
            break
    key = (ctypes.c_uint32 * len(words))(*words)
    lib().mt_init_by_array(ctypes.byref(s), key, len(words))
    return s


def sample_seq(np_state, py_state, episode_len, batch, act_len):
    el = np.ascontiguousarray(episode_len, dtype=np.int32)
    ep = np.zeros(batch, dtype=np.int64)
    st = np.zeros(batch, dtype=np.int64)
    rc = lib().replay_sample_seq(ctypes.byref(np_state), ctypes.byref(py_state), el.ctypes.data, len(el), batch,
                                 act_len, ep.ctypes.data, st.ctypes.data)
    if rc:
        raise AssertionError("episode shorter than act_len+1")
    return ep, st


def sample_mixed(np_state, py_state, len_rand, len_vid, batch, act_len, rand_prob):
    lr = np.ascontiguousarray(len_rand, dtype=np.int32)
    lv = np.ascontiguousarray(len_vid, dtype=np.int32)
    ep = np.zeros(batch, dtype=np.int64)
    st = np.zeros(batch, dtype=np.int64)
    n = lib().replay_sample_mixed(ctypes.byref(np_state), ctypes.byref(py_state), lr.ctypes.data, len(lr),
                                  lv.ctypes.data, len(lv), batch, act_len, float(rand_prob), ep.ctypes.data,
                                  st.ctypes.data)
    if n < 0:
        raise AssertionError("episode shorter than act_len+1")
    return n, ep, st
