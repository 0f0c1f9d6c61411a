"""ORACLE (test infrastructure only) -- the input-preparation step in front of the hot path.

Reference call sites (relative to /root/reference/): ``falcon-r1cs/src/circuits/falcon_ntt.rs:27-28``
(``Polynomial::from(&self.sig)``, ``Polynomial::from(&self.pk)``) and ``:44``
(``Polynomial::from_hash_of_message(msg, sig.nonce())``); same in ``examples/pok_sig.rs:33-36``.

The arithmetic lives in ``falcon-rust`` (git https://github.com/zhenfeizhang/falcon.rs, no rev pinned,
``falcon-r1cs/Cargo.toml:11``), a wrapper of the Falcon round-3 C implementation; it is NOT under /root/reference.
PARITY UNPINNED: restated from the Falcon specification (v1.2, sections 3.7 "Hashing", 3.11.2-3.11.4 "Encoding");
the reference holds no encoded key, signature or hash vector to pin it.  SHAKE256 itself is ``hashlib``'s
(FIPS 202), independent of the HIP Keccak it checks.

* public key  = header byte 0x00 + logn, then N coefficients of 14 bits each, big-endian bit order (modq_encode)
* signature   = header byte 0x30 + logn, 40-byte nonce, compressed s2 (comp_encode: sign bit, 7 low bits, high
                bits in unary 0...01), zero padded to SIG_LEN (falcon.rs uses the padded lengths 666 / 1280)
* hash_to_point(nonce || msg): SHAKE256 stream read as big-endian 16-bit words w; w < 5q = 61445 is accepted as
                w mod q, until N coefficients
"""
import hashlib

Q = 12289
NONCE_LEN = 40
SIG_LEN = {9: 666, 10: 1280}        # falcon.rs SIG_LEN (FALCON_SIG_PADDED_SIZE)
PK_LEN = {9: 897, 10: 1793}         # 1 + 14 N / 8


def hash_to_point(nonce: bytes, msg: bytes, logn: int):
    """Falcon spec Algorithm 3 (HashToPoint), the variable-time form used for public data."""
    n = 1 << logn
    want = 2 * n + 512
    while True:
        stream = hashlib.shake_256(nonce + msg).digest(want)
        out = []
        for i in range(0, len(stream) - 1, 2):
            w = (stream[i] << 8) | stream[i + 1]
            if w < 5 * Q:
                out.append(w % Q)
                if len(out) == n:
                    return out
        want *= 2


def modq_encode(coeffs, logn: int) -> bytes:
    n = 1 << logn
    assert len(coeffs) == n and all(0 <= c < Q for c in coeffs)
    acc = 0
    for c in coeffs:
        acc = (acc << 14) | c
    return bytes([logn]) + acc.to_bytes(14 * n // 8, "big")


def modq_decode(data: bytes, logn: int):
    """-> list of N coefficients, or None when the encoding is invalid (wrong header/length, coefficient >= q)."""
    n = 1 << logn
    if len(data) != PK_LEN[logn] or data[0] != logn:
        return None
    acc = int.from_bytes(data[1:], "big")
    out = [(acc >> (14 * (n - 1 - i))) & 0x3FFF for i in range(n)]
    return None if any(c >= Q for c in out) else out


def comp_encode(signed_coeffs, logn: int, nonce: bytes, sig_len=None) -> bytes:
    """Signed s2 coefficients (|x| <= 2047) -> padded signature bytes."""
    n = 1 << logn
    sig_len = sig_len or SIG_LEN[logn]
    assert len(signed_coeffs) == n and len(nonce) == NONCE_LEN
    bits = []
    for x in signed_coeffs:
        assert -2047 <= x <= 2047
        m = abs(x)
        bits.append(1 if x < 0 else 0)
        bits.extend((m >> i) & 1 for i in range(6, -1, -1))
        bits.extend([0] * (m >> 7))
        bits.append(1)
    body_len = sig_len - 1 - NONCE_LEN
    if len(bits) > 8 * body_len:
        raise ValueError("signature does not fit the padded length")
    bits.extend([0] * (8 * body_len - len(bits)))
    body = bytes(int("".join(map(str, bits[i:i + 8])), 2) for i in range(0, len(bits), 8))
    return bytes([0x30 + logn]) + nonce + body


def comp_decode(data: bytes, logn: int):
    """-> (nonce, coefficients mod q) or None.  Falcon spec Algorithm 18 (Decompress) + the codec's strictness
    rules: no "-0", |x| <= 2047, all unused trailing bits zero."""
    n = 1 << logn
    if len(data) < 1 + NONCE_LEN or data[0] != 0x30 + logn:
        return None
    nonce = data[1:1 + NONCE_LEN]
    body = data[1 + NONCE_LEN:]
    total = 8 * len(body)
    bit = lambda i: (body[i >> 3] >> (7 - (i & 7))) & 1
    pos = 0
    out = []
    for _ in range(n):
        if pos + 8 > total:
            return None
        s = bit(pos)
        m = 0
        for k in range(1, 8):
            m = (m << 1) | bit(pos + k)
        pos += 8
        while True:
            if pos >= total:
                return None
            b = bit(pos)
            pos += 1
            if b:
                break
            m += 128
            if m > 2047:
                return None
        if s and m == 0:
            return None
        out.append((Q - m) if s else m)
    if any(bit(i) for i in range(pos, total)):
        return None
    return nonce, out


def signed_to_modq(x):
    return x % Q
