"""Integer known-answer tests for the oracle's RNG and time schedule.

The reference holds no fixtures, so the answers are derived here a second time,
independently of oracle/rt_oracle.c, straight from the formulas at
reference src/shaders/raytrace_kernal.wgsl:255-275 in numpy uint32 arithmetic, and cross-checked
against the surveyor's table.
"""
import numpy as np

import oracle_lib as O

M32 = 0xFFFFFFFF


def py_seed(u, v, tbits):
    x = ((u << 15) & M32) ^ ((u * 1729 + 8192374) & M32)
    y = ((v << 21) & M32) ^ ((v * 271 + 719827371) & M32)
    t = (tbits * 87189 + 189273) & M32
    return x ^ y ^ t


def py_step(r):
    r = (r * 18197 + 182489729) & M32
    r ^= (r << 15) & M32
    r ^= r >> 22
    r ^= (r << 4) & M32
    return r


def py_float(r):
    return float(np.array([(r & 0x007FFFFF) | 0x3F800000], np.uint32).view(np.float32)[0] - np.float32(1.0))


def py_time_bits(n_additions):
    t = np.float32(0.0)
    for _ in range(n_additions):
        t = np.float32(t + np.float32(0.01))
    return int(np.array([t], np.float32).view(np.uint32)[0])


# (u, v, frame) -> time bits, seed, state after the discarded rand, next three states, as floats — SURVEY.md §8c
SURVEY_KATS = [
    ((0, 0, 0), 0x3C23D70A, 0xB0E2E4F6, 0xBA0B70BD, (0xADD7D790, 0x1D9DA83A, 0x05B98160), (0.68626595, 0.23169637, 0.4492607)),
    ((1, 2, 0), 0x3C23D70A, 0xB0A26FD5, 0xA167D0DD, (0x248F3D10, 0xA970FB6A, 0x26A7BB98), (0.11905098, 0.88267255, 0.3104124)),
    ((639, 351, 0), 0x3C23D70A, 0x9AC32C22, 0x89D4CDDD, (0x5DF0E7FA, 0x087F6C1C, 0xED5DFBA5), (0.88207936, 0.99548674, 0.7342421)),
    ((1919, 1071, 3), 0x3D23D70A, 0x83646972, 0x43532A6B, (0x39792229, 0xD74D706A, 0x59EAAF68), (0.946355, 0.6049931, 0.833478)),
    ((3839, 2159, 63), 0x3F23D705, 0x79FCFBDB, 0xEE296C41, (0xEE50DC06, 0x2F8B4FBC, 0x75C1C3ED), (0.6317146, 0.0883708, 0.5137917)),
]


def test_time_bits_by_repeated_addition():
    lib = O.load()
    # frame f of a fresh process runs with f+1 additions of 0.01f (app.rs:94 before the first render)
    expect = {0: 0x3C23D70A, 1: 0x3CA3D70A, 2: 0x3CF5C28F, 3: 0x3D23D70A, 4: 0x3D4CCCCC, 63: 0x3F23D705, 255: 0x4023D702}
    for f, bits in expect.items():
        assert py_time_bits(f + 1) == bits
        assert lib.rto_time_advance(0, f + 1) == bits
    table = O.time_bits_table(256)
    for f, bits in expect.items():
        assert int(table[f]) == bits
    # frame 63 is NOT round(64 * 0.01f): the schedule must be produced by repeated addition
    assert int(np.array([np.float32(64) * np.float32(0.01)], np.float32).view(np.uint32)[0]) != expect[63]


def test_survey_kats_independent_derivation_and_oracle():
    import ctypes as C

    lib = O.load()
    for (u, v, frame), tbits, seed, after_discard, nxt, floats in SURVEY_KATS:
        assert py_time_bits(frame + 1) == tbits
        assert py_seed(u, v, tbits) == seed
        assert lib.rto_rng_seed(u, v, tbits) == seed
        s = py_step(seed)
        assert s == after_discard
        so = lib.rto_rng_step(seed, None)
        assert so == after_discard
        for want_state, want_float in zip(nxt, floats):
            s = py_step(s)
            f = C.c_float()
            so = lib.rto_rng_step(so, C.byref(f))
            assert s == want_state and so == want_state
            assert abs(py_float(s) - want_float) < 1e-7
            assert np.float32(f.value) == np.float32(py_float(s))  # exact: (bits | 0x3f800000) - 1.0


def test_rng_stream_long_run_matches_numpy():
    import ctypes as C

    lib = O.load()
    rs = np.random.RandomState(7)
    for _ in range(20):
        u, v, tb = (int(x) for x in rs.randint(0, 2**31, 3))
        s_py = py_seed(u & 0xFFFF, v & 0xFFFF, tb)
        s_c = lib.rto_rng_seed(u & 0xFFFF, v & 0xFFFF, tb)
        assert s_py == s_c
        f = C.c_float()
        for _ in range(200):
            s_py = py_step(s_py)
            s_c = lib.rto_rng_step(s_c, C.byref(f))
            assert s_py == s_c
            assert 0.0 <= f.value < 1.0


def test_rand_has_23_bit_resolution_and_unit_range():
    # lowest / highest mantissa map to 0 and 1 - 2^-23 (wgsl:272-274)
    assert py_float(0x00000000) == 0.0
    assert py_float(0x007FFFFF) == 1.0 - 2.0**-23
