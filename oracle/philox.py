"""TEST INFRASTRUCTURE (oracle) -- never imported by the product path.

CPU restatement of the counter-based random streams the device kernels use (rlcontrol_amd/csrc/rlc_common.h):
Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11), the
`sample_n_k` replacement `rlc_sample_distinct` (k distinct uniform indices in range(n), the device counterpart of
utils/custom_collections.py:107-131 of the reference), the Box-Muller normals of the device OU process
(utils/exploration_policy.py:18-21) and the uniform draws of the device environment reset.

Pinned by the Random123 known-answer vectors (tests/test_philox.py); the device kernels are compared against this
file bit for bit (integers) / to float rounding (normals).
"""
import numpy as np

M0 = 0xD2511F53
M1 = 0xCD9E8D57
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK32 = 0xFFFFFFFF
MASK64 = 0xFFFFFFFFFFFFFFFF

# stream separators xor-ed into the per-agent seed (must match the kernels)
KEY_OU = 0x5DEECE66D            # ddpg act kernel / train-step kernel: OU normals, counter = draws so far
KEY_ENV_TRAIN = 0x7261696E      # train-environment reset stream, counter = resets so far
KEY_ENV_TEST = 0x74657374       # test-environment reset stream, counter = eval_round*eval_episodes + episode


def philox4x32_10(key, ctr_lo, ctr_hi):
    """key, ctr_lo, ctr_hi: 64-bit ints. Returns the four 32-bit output words (x, y, z, w)."""
    k0, k1 = key & MASK32, (key >> 32) & MASK32
    c0, c1 = ctr_lo & MASK32, (ctr_lo >> 32) & MASK32
    c2, c3 = ctr_hi & MASK32, (ctr_hi >> 32) & MASK32
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        n0 = ((p1 >> 32) ^ c1 ^ k0) & MASK32
        n1 = p1 & MASK32
        n2 = ((p0 >> 32) ^ c3 ^ k1) & MASK32
        n3 = p0 & MASK32
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + W0) & MASK32
        k1 = (k1 + W1) & MASK32
    return c0, c1, c2, c3


def below(p, n):
    """integer in [0, n): high 64 bits of (64 random bits * n)  (philox_below)"""
    u = (p[0] << 32) | p[1]
    return (u * n) >> 64


def sample_distinct(n, k, key, call):
    """k distinct logical indices in range(n) (rlc_sample_distinct): both regimes."""
    if 3 * k >= n:
        pool = list(range(n))
        for i in range(k):
            p = philox4x32_10(key, call, 0x100000000 + i)
            j = i + below(p, n - i)
            pool[i], pool[j] = pool[j], pool[i]
        return np.array(pool[:k], dtype=np.int64)
    out = [-1] * k
    need = [True] * k
    rnd = 0
    while True:
        for t in range(k):
            if need[t]:
                out[t] = below(philox4x32_10(key, call, (rnd << 32) | t), n)
        need = [any(out[j] == out[t] for j in range(t)) for t in range(k)]
        if not any(need):
            return np.array(out, dtype=np.int64)
        rnd += 1


def normal2(p):
    """two standard normals (float32 arithmetic, Box-Muller on (0,1] x [0,1))  (philox_normal2)"""
    f = np.float32
    u0 = (f(p[0] >> 8) + f(1.0)) * f(1.0 / 16777216.0)
    u1 = f(p[1] >> 8) * f(1.0 / 16777216.0)
    rad = np.sqrt(f(-2.0) * np.log(u0, dtype=np.float32), dtype=np.float32)
    ang = f(6.28318530717958647692) * u1
    return f(rad * np.cos(ang, dtype=np.float32)), f(rad * np.sin(ang, dtype=np.float32))


def uniform01_double(word_hi, word_lo):
    """double in [0,1) from 53 random bits (device environment reset)"""
    bits = ((word_hi << 32) | word_lo) >> 11
    return bits * (1.0 / 9007199254740992.0)
