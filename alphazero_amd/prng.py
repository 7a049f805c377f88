"""Host-side key handling for the JAX-style threefry generator.

Keys are two uint32 words kept on the host (they parameterise kernel launches;
the random bits themselves are generated on the device).  Key derivation uses
the "partitionable" counter layout of JAX >= 0.5: child i of `split` is the
threefry-2x32 block of counter (0, i), both output words kept.
"""
import numpy as np

_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))
_M32 = 0xFFFFFFFF


def _rotl(v, r):
    return ((v << r) | (v >> (32 - r))) & _M32


def threefry2x32(key, ctr):
    """Threefry-2x32, 20 rounds (Salmon et al. 2011) on Python ints."""
    k0, k1 = int(key[0]) & _M32, int(key[1]) & _M32
    ks = (k0, k1, k0 ^ k1 ^ 0x1BD11BDA)
    x0 = (int(ctr[0]) + ks[0]) & _M32
    x1 = (int(ctr[1]) + ks[1]) & _M32
    for block in range(5):
        for r in _ROT[block % 2]:
            x0 = (x0 + x1) & _M32
            x1 = _rotl(x1, r) ^ x0
        x0 = (x0 + ks[(block + 1) % 3]) & _M32
        x1 = (x1 + ks[(block + 2) % 3] + block + 1) & _M32
    return x0, x1


def as_key(rng_key):
    """Accepts an int seed (PRNGKey(seed) = [0, seed]), a pair of words, a numpy array or a
    2-element tensor; returns a tuple of two Python ints."""
    if isinstance(rng_key, (int, np.integer)):
        seed = int(rng_key)
        return ((seed >> 32) & _M32, seed & _M32)
    if hasattr(rng_key, "detach"):  # torch tensor
        rng_key = rng_key.detach().cpu().numpy()
    arr = np.asarray(rng_key).reshape(-1)
    if arr.size != 2:
        raise ValueError("rng_key must be an int seed or two 32-bit words")
    return (int(arr[0]) & _M32, int(arr[1]) & _M32)


def split(rng_key, num=2):
    """jax.random.split in the partitionable layout: list of `num` keys."""
    key = as_key(rng_key)
    return [threefry2x32(key, (0, i)) for i in range(num)]


def key_array(key):
    return (np.uint32(key[0]), np.uint32(key[1]))


def random_bits32(key, n):
    """n words of jax.random.bits(key, (n,), uint32) in the partitionable layout: word i is
    the xor of the two outputs of the threefry-2x32 block with counter (0, i).  NumPy, vectorised
    over i (the device counterpart is az_random_bits32 in csrc/az_math.h)."""
    k0, k1 = (np.uint32(w) for w in as_key(key))
    ks = (k0, k1, np.uint32(k0 ^ k1 ^ np.uint32(0x1BD11BDA)))
    with np.errstate(over="ignore"):
        x0 = np.zeros(n, np.uint32) + ks[0]
        x1 = np.arange(n, dtype=np.uint32) + ks[1]
        for block in range(5):
            for r in _ROT[block % 2]:
                x0 = x0 + x1
                x1 = ((x1 << np.uint32(r)) | (x1 >> np.uint32(32 - r))) ^ x0
            x0 = x0 + ks[(block + 1) % 3]
            x1 = x1 + ks[(block + 2) % 3] + np.uint32(block + 1)
    return x0 ^ x1


def permutation(key, n):
    """jax.random.permutation(key, n) -- and, the keys depending on n alone, the order in which
    jax.random.permutation(key, x) leaves the rows of any x with n rows -- as jax._src.random._shuffle
    does it: ceil(3 ln n / ln(2^32 - 1)) rounds of `key, subkey = split(key)`, 32 random bits per
    element from subkey, stable sort by those bits.  Returns int64 indices (NumPy)."""
    n = int(n)
    x = np.arange(n, dtype=np.int64)
    rounds = int(np.ceil(3 * np.log(max(1, n)) / np.log(np.iinfo(np.uint32).max)))
    key = as_key(key)
    for _ in range(rounds):
        key, subkey = split(key, 2)
        x = x[np.argsort(random_bits32(subkey, n), kind="stable")]
    return x
