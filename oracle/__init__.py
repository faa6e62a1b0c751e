"""CPU ORACLE -- test infrastructure, not product code.

numpy-facing wrapper around ``oracle/stereo_oracle.c`` (a plain-C restatement
of the reference's cost-volume / disparity-regression path, each function
citing the reference file:line it follows).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; ``stereoformer_amd`` never does.

Pinned against golden tensors generated from the real reference
(``tests/golden/make_golden.py``); ``cross_attention`` is parity-unpinned by
the reference (its file is empty) and pinned by identities only.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libstereo_oracle.so")
_lib = None

_f = ctypes.POINTER(ctypes.c_float)
_i = ctypes.c_int
_ll = ctypes.c_longlong


def build(force: bool = False, native: bool = False, out: str | None = None) -> str:
    """Compile the C oracle with gcc.  ``native`` adds -march=native (bench
    cpu_baseline leg on the GPU host); the default build is portable."""
    target = out or _LIB_PATH
    src = os.path.join(_HERE, "stereo_oracle.c")
    if not force and os.path.exists(target) and os.path.getmtime(target) >= os.path.getmtime(src):
        return target
    march = "native" if native else "x86-64-v2"
    cmd = ["gcc", "-O3", f"-march={march}", "-fopenmp", "-fPIC", "-std=c11", "-fno-fast-math",
           "-ffp-contract=off", "-shared", "-o", target, src, "-lm"]
    subprocess.run(cmd, check=True)
    return target


def load(path: str | None = None):
    global _lib
    if path is None and _lib is not None:
        return _lib
    p = path or _LIB_PATH
    if not os.path.exists(p):
        build()
    lib = ctypes.CDLL(p)
    lib.oracle_gwc_volume.restype = ctypes.c_int
    if path is None:
        _lib = lib
    return lib


def set_threads(n: int, lib=None) -> None:
    """OpenMP thread count for the following oracle calls (cpu_baseline reports the count it used)."""
    (lib or load()).oracle_set_threads(_i(int(n)))


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(_f)


# --------------------------------------------------------------- volumes
def corr_volume(L, R, D, lib=None):
    L, R = _c(L), _c(R)
    B, C, H, W = L.shape
    out = np.empty((B, D, H, W), np.float32)
    (lib or load()).oracle_corr_volume(_p(L), _p(R), _p(out), _i(B), _i(C), _i(H), _i(W), _i(D))
    return out


def diff_volume(L, R, D):
    L, R = _c(L), _c(R)
    B, C, H, W = L.shape
    out = np.empty((B, C, D, H, W), np.float32)
    load().oracle_diff_volume(_p(L), _p(R), _p(out), _i(B), _i(C), _i(H), _i(W), _i(D))
    return out


def concat_volume(L, R, D):
    L, R = _c(L), _c(R)
    B, C, H, W = L.shape
    out = np.empty((B, 2 * C, D, H, W), np.float32)
    load().oracle_concat_volume(_p(L), _p(R), _p(out), _i(B), _i(C), _i(H), _i(W), _i(D))
    return out


def gwc_volume(L, R, D, G, lib=None):
    L, R = _c(L), _c(R)
    B, C, H, W = L.shape
    out = np.empty((B, G, D, H, W), np.float32)
    rc = (lib or load()).oracle_gwc_volume(_p(L), _p(R), _p(out), _i(B), _i(C), _i(H), _i(W), _i(D), _i(G))
    if rc != 0:
        raise AssertionError("C % num_groups != 0")  # GWC_Cost_Volume.py:7
    return out


def group_mean(volg, lib=None):
    volg = _c(volg)
    B, G, D, H, W = volg.shape
    out = np.empty((B, D, H, W), np.float32)
    (lib or load()).oracle_group_mean(_p(volg), _p(out), _i(B), _i(G), _i(D), _i(H), _i(W))
    return out


# --------------------------------------------------------------- regression
def soft_argmin(vol, match_similarity=True, with_prob=False, lib=None):
    vol = _c(vol)
    assert vol.ndim == 4  # estimation.py:15
    B, D, H, W = vol.shape
    disp = np.empty((B, H, W), np.float32)
    prob = np.empty_like(vol) if with_prob else None
    (lib or load()).oracle_soft_argmin(_p(vol), _p(disp), _p(prob) if with_prob else None,
                                       _i(B), _i(D), _i(H), _i(W), _i(1 if match_similarity else 0))
    return (disp, prob) if with_prob else disp


def sample_volume(vol, cand):
    vol, cand = _c(vol), _c(cand)
    B, D, H, W = vol.shape
    K = cand.shape[1]
    out = np.empty_like(cand)
    load().oracle_sample_volume(_p(vol), _p(cand), _p(out), _i(B), _i(D), _i(H), _i(W), _i(K))
    return out


def local_lookup(vol, disp, radius, S, consider_valid=False, lib=None):
    vol, disp = _c(vol), _c(disp)
    B, D, H, W = vol.shape
    out = np.empty((B, 1, H, W), np.float32)
    (lib or load()).oracle_local_lookup(_p(vol), _p(disp), _p(out), _i(B), _i(D), _i(H), _i(W),
                                        ctypes.c_float(radius), _i(S), _i(int(consider_valid)))
    return out


def variance_lookup(vol, disp, gamma, S, consider_valid=False):
    vol, disp = _c(vol), _c(disp)
    B, D, H, W = vol.shape
    out = np.empty((B, 1, H, W), np.float32)
    load().oracle_variance_lookup(_p(vol), _p(disp), _p(out), _i(B), _i(D), _i(H), _i(W),
                                  ctypes.c_float(gamma), _i(S), _i(int(consider_valid)))
    return out


def bounds_lookup(vol, disp, lower_off, upper_off, S, consider_valid=True):
    vol, disp, lo, up = _c(vol), _c(disp), _c(lower_off), _c(upper_off)
    B, D, H, W = vol.shape
    out = np.empty((B, 1, H, W), np.float32)
    load().oracle_bounds_lookup(_p(vol), _p(disp), _p(lo), _p(up), _p(out), _i(B), _i(D), _i(H), _i(W),
                                _i(S), _i(int(consider_valid)))
    return out


def uncertainty_volume(vol, disp):
    vol, disp = _c(vol), _c(disp)
    B, D, H, W = vol.shape
    out = np.empty_like(vol)
    load().oracle_uncertainty_volume(_p(vol), _p(disp), _p(out), _i(B), _i(D), _i(H), _i(W))
    return out


# --------------------------------------------------------------- RAFT path
def allpairs_corr(f1, f2):
    f1, f2 = _c(f1), _c(f2)
    B, C, H, W1 = f1.shape
    W2 = f2.shape[3]
    out = np.empty((B, H, W1, W2), np.float32)
    load().oracle_allpairs_corr(_p(f1), _p(f2), _p(out), _i(B), _i(C), _i(H), _i(W1), _i(W2))
    return out


def corr_pyramid(corr, num_levels):
    """[B,H,W1,W2] -> list of num_levels+1 arrays [B*H*W1, W2/2^i] (corr.py:66-76)."""
    B, H, W1, W2 = corr.shape
    lv = [_c(corr).reshape(B * H * W1, W2)]
    for _ in range(num_levels):
        prev = lv[-1]
        nxt = np.empty((prev.shape[0], prev.shape[1] // 2), np.float32)
        load().oracle_avgpool_w2(_p(prev), _p(nxt), _ll(prev.shape[0]), _i(prev.shape[1]))
        lv.append(nxt)
    return lv


def corr_lookup(pyramid, coords, radius, num_levels):
    coords = _c(coords)
    B, _, H, W1 = coords.shape
    L = num_levels
    T = 2 * radius + 1
    out = np.empty((B, L * T, H, W1), np.float32)
    lv = [_c(a) for a in pyramid[:L]]
    ptrs = (_f * L)(*[_p(a) for a in lv])
    widths = (ctypes.c_int * L)(*[a.shape[1] for a in lv])
    load().oracle_corr_lookup(ptrs, widths, _i(L), _i(radius), _p(coords), _p(out), _i(B), _i(H), _i(W1))
    return out


def alt_corr_lookup(f1, f2, coords, radius, num_levels):
    f1, f2, coords = _c(f1), _c(f2), _c(coords)
    B, C, H, W = f1.shape
    T = 2 * radius + 1
    out = np.empty((B, num_levels * T, H, W), np.float32)
    load().oracle_alt_corr_lookup(_p(f1), _p(f2), _p(coords), _p(out), _i(B), _i(C), _i(H), _i(W),
                                  _i(num_levels), _i(radius))
    return out


# --------------------------------------------------------------- warp / upsample
def disp_warp(img, disp, padding_mode="border"):
    img, disp = _c(img), _c(disp)
    assert disp.min() >= 0  # disparity_warper.py:93
    B, Ci, H, W = img.shape
    warped = np.empty_like(img)
    mask = np.empty_like(img)
    load().oracle_disp_warp(_p(img), _p(disp), _p(warped), _p(mask), _i(B), _i(Ci), _i(H), _i(W),
                            _i(1 if padding_mode == "border" else 0))
    return warped, mask


def upsample_convex(disp, mask, factor):
    disp, mask = _c(disp), _c(mask)
    N, Cd, H, W = disp.shape
    out = np.empty((N, Cd, factor * H, factor * W), np.float32)
    load().oracle_upsample_convex(_p(disp), _p(mask), _p(out), _i(N), _i(Cd), _i(H), _i(W), _i(factor))
    return out


def resize_bilinear(x, size, align_corners):
    x = _c(x)
    N, C, Hi, Wi = x.shape
    Ho, Wo = size
    out = np.empty((N, C, Ho, Wo), np.float32)
    load().oracle_resize_bilinear(_p(x), _p(out), _i(N * C), _i(Hi), _i(Wi), _i(Ho), _i(Wo),
                                  _i(int(align_corners)))
    return out


def upsample_simple(disp, factor):
    """upsample_simple8 / upflow8: factor * bilinear(align_corners=True)."""
    N, C, H, W = disp.shape
    return (factor * resize_bilinear(disp, (factor * H, factor * W), True)).astype(np.float32)


# --------------------------------------------------------------- attention (unpinned)
def cross_attention(L, R, V, D, mask="zero", scale="sqrt_c", lib=None):
    L, R = _c(L), _c(R)
    V = R if V is None else _c(V)
    B, C, H, W = L.shape
    Cv = V.shape[1]
    out = np.empty((B, Cv, H, W), np.float32)
    disp = np.empty((B, H, W), np.float32)
    (lib or load()).oracle_cross_attention(_p(L), _p(R), _p(V), _p(out), _p(disp), _i(B), _i(C), _i(Cv), _i(H),
                                           _i(W), _i(D), _i({"zero": 0, "neginf": 1}[mask]),
                                           _i({"sqrt_c": 0, "mean": 1}[scale]))
    return out, disp


# --------------------------------------------------------------- GRU pointwise
def gru_blend(h, zc, qc, cz=None, cq=None, softsign=False):
    h, zc, qc = _c(h), _c(zc), _c(qc)
    cz = None if cz is None else _c(cz)
    cq = None if cq is None else _c(cq)
    out = np.empty_like(h)
    load().oracle_gru_blend(_p(h), _p(zc), _p(qc), _p(cz) if cz is not None else None,
                            _p(cq) if cq is not None else None, _p(out), _ll(h.size), _i(int(softsign)))
    return out


def gru_reset(h, rc, cr=None):
    h, rc = _c(h), _c(rc)
    cr = None if cr is None else _c(cr)
    out = np.empty_like(h)
    load().oracle_gru_reset(_p(h), _p(rc), _p(cr) if cr is not None else None, _p(out), _ll(h.size))
    return out


# --------------------------------------------------------------- extractor epilogue (N2)
def instance_norm_relu(x, gamma=None, beta=None, eps=1e-5, relu=True):
    x = _c(x)
    B, C, H, W = x.shape
    g = None if gamma is None else _c(gamma)
    b = None if beta is None else _c(beta)
    out = np.empty_like(x)
    load().oracle_instance_norm_relu(_p(x), _p(g) if g is not None else None, _p(b) if b is not None else None, _p(out),
                                     _i(B), _i(C), _i(H), _i(W), ctypes.c_float(eps), _i(int(relu)))
    return out
