"""Host-side driver of the witness engine (Python face of the C ABI in include/frw.h).

Mirrors what a caller of the reference does around
``FalconNTTVerificationCircuit::generate_constraints`` (falcon-r1cs/src/circuits/falcon_ntt.rs:26-123):
hand over (sig, pk, hm) coefficient vectors, receive ``witness_assignment`` / ``instance_assignment``
of every signature in arkworks order.  torch is used only to own device memory and streams.
"""
import ctypes as C
import weakref
from dataclasses import dataclass

import numpy as np

from ._lib import CompactLayoutStruct, FrwError, LayoutDualStruct, LayoutStruct, check, load_library

ENC_CANONICAL, ENC_MONTGOMERY, ENC_COMPACT = 0, 1, 2
ST_OK, ST_COEFF_RANGE, ST_NORM_BOUND, ST_DECODE = 0, 1, 2, 3
NONCE_LEN = 40
PK_LEN = {9: 897, 10: 1793}
SIG_LEN = {9: 666, 10: 1280}
E_RANGE = -5
G_LESS_THAN_Q, G_MOD_Q, G_ADD_MOD, G_L2_ELEM, G_NORM_BOUND_512, G_NORM_BOUND_1024 = range(6)


@dataclass(frozen=True)
class Layout:
    logn: int
    n: int
    num_witness: int
    num_instance: int
    num_constraints: int
    seg_off: tuple
    seg_len: tuple


def layout(logn) -> Layout:
    s = LayoutStruct()
    check(load_library().frw_layout(int(logn), C.byref(s)), "frw_layout")
    return Layout(s.logn, s.n, s.num_witness, s.num_instance, s.num_constraints, tuple(s.seg_off), tuple(s.seg_len))


def layout_dual(logn) -> Layout:
    """Layout of FalconDualNTTVerificationCircuit's witness (frw_layout_dual; 15 segments)."""
    s = LayoutDualStruct()
    check(load_library().frw_layout_dual(int(logn), C.byref(s)), "frw_layout_dual")
    return Layout(s.logn, s.n, s.num_witness, s.num_instance, s.num_constraints, tuple(s.seg_off), tuple(s.seg_len))


@dataclass(frozen=True)
class CompactLayout:
    logn: int
    n: int
    bytes_per_signature: int
    small_off: int
    num_small: int
    t_off: int
    num_t: int
    bits_off: int
    num_bit_words: int
    bit_seg_off: tuple
    instance_off: int
    num_instance_values: int
    status_off: int


# Launch shapes of the two launches bench.py times on an MI355X (256 CUs, 3 resident workgroups per CU, 4x
# over-subscription): (logn, signatures per launch) -> (grid, signatures beyond the last full round, cut into five work
# items each).  bench.py asserts its timed launches against this table and tests/test_gpu_parity.py asserts the table
# against the library, so the benchmark, its test and their descriptions cannot drift apart again.
MI355X_BENCH_LAUNCH_SHAPES = {(10, 32768): (3072, 2048), (9, 8192): (2048, 0)}


def compact_layout(logn) -> CompactLayout:
    """Layout of FRW_ENC_COMPACT (frw_compact_layout): 11N u32 values + 2N five-limb quotients + boolean bit array +
    2N u32 instance values."""
    s = CompactLayoutStruct()
    check(load_library().frw_compact_layout(int(logn), C.byref(s)), "frw_compact_layout")
    return CompactLayout(s.logn, s.n, s.bytes_per_signature, s.small_off, s.num_small, s.t_off, s.num_t, s.bits_off, s.num_bit_words,
                         tuple(s.bit_seg_off), s.instance_off, s.num_instance_values, s.status_off)


def synth_triples(logn, batch, seed=0x46414C434F4E, first_index=0):
    """Synthetic valid (sig, pk, hm) -- frw_synth_triples; uint16 arrays of shape [batch, N]."""
    n = 1 << logn
    out = [np.empty((batch, n), dtype=np.uint16) for _ in range(3)]
    check(load_library().frw_synth_triples(int(logn), batch, seed, first_index,
                                           *[a.ctypes.data_as(C.c_void_p) for a in out]), "frw_synth_triples")
    return tuple(out)


VERIFY_POINTS_ARE_CHECKED = 1
VK_POINTS_ARE_CHECKED = 1
KEY_AUTO, KEY_TABLES, KEY_BARE = 0, 1, 2           # frw.h FRW_KEY_*: window tables, or the points only (keys that would not fit)
GROTH16_PARTIAL_WORDS = 72
GROTH16_COMBINE_WORKSPACE = 4096


class Groth16Verifier:
    """ark-groth16's prepared verifying key + verify_proof (examples/pok_sig.rs:34-47).  Host code: no device involved.

    vk: the dict WitnessEngine.groth16_setup returns, or one flat uint64 array in frw_groth16_setup's vk_out layout."""

    def __init__(self, vk, points_are_checked=False):
        self._lib = load_library()
        if isinstance(vk, dict):
            vk = np.concatenate([np.asarray(vk[k], dtype=np.uint64).reshape(-1) for k in ("alpha_g1", "beta_g2", "gamma_g2", "delta_g2", "gamma_abc_g1")])
        vk = np.ascontiguousarray(vk, dtype=np.uint64).reshape(-1)
        if vk.size < 96 or (vk.size - 84) % 12:
            raise FrwError("verifying key: expected 84 + 12 x num_instance uint64 values")
        self.num_instance = (vk.size - 84) // 12
        self._h = C.c_void_p()
        check(self._lib.frw_groth16_vk_load_opts(vk.ctypes.data_as(C.c_void_p), self.num_instance, VK_POINTS_ARE_CHECKED if points_are_checked else 0,
                                                 C.byref(self._h)), "frw_groth16_vk_load")

    def verify(self, instance, proofs, encoding=ENC_MONTGOMERY, flags=0):
        """instance: uint64[batch, num_instance, 4] as the witness entry points write it (the constant one first);
        proofs: uint64[batch, 48].  Returns int32[batch]: 1 accepted, 0 rejected, -1 malformed."""
        proofs = np.ascontiguousarray(proofs).view(np.uint64).reshape(-1, 48)
        instance = np.ascontiguousarray(instance).view(np.uint64).reshape(proofs.shape[0], self.num_instance, 4)
        out = np.zeros(proofs.shape[0], dtype=np.int32)
        check(self._lib.frw_groth16_verify(self._h, proofs.shape[0], instance.ctypes.data_as(C.c_void_p), int(encoding),
                                           proofs.ctypes.data_as(C.c_void_p), int(flags), out.ctypes.data_as(C.c_void_p)), "frw_groth16_verify")
        return out

    def close(self):
        if self._h:
            self._lib.frw_groth16_vk_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def diag_pairing(g1, g2):
    """The verifier's pairing of one pair (ark-ff limbs: 12 and 24 uint64): uint64[12, 6], see frw.h."""
    g1 = np.ascontiguousarray(g1, dtype=np.uint64).reshape(12)
    g2 = np.ascontiguousarray(g2, dtype=np.uint64).reshape(24)
    out = np.zeros((12, 6), dtype=np.uint64)
    check(load_library().frw_diag_pairing(g1.ctypes.data_as(C.c_void_p), g2.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)), "frw_diag_pairing")
    return out


def _u16(a, n):
    a = np.ascontiguousarray(a, dtype=np.uint16)
    if a.ndim == 1:
        a = a.reshape(1, -1)
    if a.ndim != 2 or a.shape[1] != n:
        raise ValueError("input length %s is not N=%d" % (a.shape, n))   # poly.rs:110-112 panics likewise
    return a


class WitnessEngine:
    """One context on one HIP device.  Raises FrwError when no device is usable."""

    def __init__(self, device=0):
        self._lib = load_library()
        h = C.c_void_p()
        check(self._lib.frw_ctx_create(int(device), C.byref(h)), "frw_ctx_create")
        self._ctx = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.frw_ctx_destroy(self._ctx)
            self._ctx = None

    __del__ = close

    def host_allocations(self):
        """Device / page-locked allocations the host-buffer entry points of this context have made so far."""
        n = C.c_uint64()
        check(self._lib.frw_diag_host_allocations(self._ctx, C.byref(n)), "frw_diag_host_allocations")
        return int(n.value)

    def valu_rates(self):
        """{v_add_u32, v_mad_u64_u32: wave-instructions / SIMD / us; f29_mul_products_per_s; simds} measured now."""
        out = (C.c_double * 4)()
        check(self._lib.frw_diag_valu_rates(self._ctx, C.byref(out)), "frw_diag_valu_rates")
        return {"v_add_u32": out[0], "v_mad_u64_u32": out[1], "f29_mul_products_per_s": out[2], "simds": int(out[3])}

    def trim(self):
        """Gives the working memory of the host-buffer entry points back to the device (the next call allocates again)."""
        check(self._lib.frw_ctx_trim(self._ctx), "frw_ctx_trim")

    def pinned_empty(self, shape, dtype):
        """numpy array over page-locked host memory (frw_host_alloc); freed when the array is collected."""
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        ptr = C.c_void_p()
        check(self._lib.frw_host_alloc(self._ctx, nbytes, C.byref(ptr)), "frw_host_alloc")
        buf = (C.c_char * max(nbytes, 1)).from_address(ptr.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        lib, addr = self._lib, ptr.value
        weakref.finalize(buf, lambda: lib.frw_host_free(None, C.c_void_p(addr)))     # may outlive this engine
        return arr

    # ---- host buffers ------------------------------------------------------------------
    def witness_dual_ntt_verify(self, logn, sig, pk, hm, encoding=ENC_MONTGOMERY, strict=True):
        """FalconDualNTTVerificationCircuit (falcon_dual_ntt.rs:26-132); same conventions as witness_ntt_verify."""
        return self.witness_ntt_verify(logn, sig, pk, hm, encoding, strict, dual=True)

    def witness_ntt_verify(self, logn, sig, pk, hm, encoding=ENC_MONTGOMERY, strict=True, dual=False, pinned=False):
        """-> (witness u64[batch, W, 4], instance u64[batch, I, 4], status i32[batch]).
        pinned=True puts the outputs in page-locked memory so that the D2H copies overlap with the kernels."""
        L = layout_dual(logn) if dual else layout(logn)
        sig, pk, hm = (_u16(a, L.n) for a in (sig, pk, hm))
        batch = sig.shape[0]
        if pk.shape[0] != batch or hm.shape[0] != batch:
            raise ValueError("batch mismatch")
        if pinned:
            wit = self.pinned_empty((batch, L.num_witness, 4), np.uint64)
            inst = self.pinned_empty((batch, L.num_instance, 4), np.uint64)
            st = self.pinned_empty((batch,), np.int32)
            st[:] = 0
        else:
            wit = np.zeros((batch, L.num_witness, 4), dtype=np.uint64)
            inst = np.zeros((batch, L.num_instance, 4), dtype=np.uint64)
            st = np.zeros(batch, dtype=np.int32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        fn = self._lib.frw_witness_dual_ntt_verify if dual else self._lib.frw_witness_ntt_verify
        rc = fn(self._ctx, logn, batch, p(sig), p(pk), p(hm), encoding, p(wit), p(inst), p(st), 1 if strict else 0)
        if rc == E_RANGE:
            bad = np.nonzero(st)[0]
            raise FrwError(rc, "frw_witness_ntt_verify",
                           "Invalid input: signature(s) %s failed range checks (status %s)" % (bad[:8], st[bad][:8]))
        check(rc, "frw_witness_ntt_verify")
        return wit, inst, st

    def witness_ntt_verify_compact(self, logn, sig, pk, hm, strict=True, pinned=False):
        """Host-buffer entry point with FRW_ENC_COMPACT: -> (compact u8[batch, bytes_per_signature], status)."""
        CL = compact_layout(logn)
        sig, pk, hm = (_u16(a, CL.n) for a in (sig, pk, hm))
        batch = sig.shape[0]
        mk = (lambda shape, dt: self.pinned_empty(shape, dt)) if pinned else (lambda shape, dt: np.zeros(shape, dtype=dt))
        comp, st = mk((batch, CL.bytes_per_signature), np.uint8), mk((batch,), np.int32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        rc = self._lib.frw_witness_ntt_verify(self._ctx, logn, batch, p(sig), p(pk), p(hm), ENC_COMPACT, p(comp), None, p(st),
                                              1 if strict else 0)
        if rc == E_RANGE:
            raise FrwError(rc, "frw_witness_ntt_verify", "Invalid input: a signature failed its range checks")
        check(rc, "frw_witness_ntt_verify")
        return comp, st

    def expand_host(self, logn, compact):
        """frw_expand_host: compact u8[batch, bytes_per_signature] -> (witness u64[batch, W, 4], instance u64[batch, I, 4])."""
        L = layout(logn)
        compact = np.ascontiguousarray(compact, dtype=np.uint8)
        batch = compact.shape[0]
        wit = np.empty((batch, L.num_witness, 4), dtype=np.uint64)
        inst = np.empty((batch, L.num_instance, 4), dtype=np.uint64)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        check(self._lib.frw_expand_host(logn, batch, p(compact), p(wit), p(inst)), "frw_expand_host")
        return wit, inst

    def aggregate(self, items, encoding=ENC_MONTGOMERY, strict=True):
        """Aggregate driver (the reference's falcon-aggregate-sig is an empty stub, so the behaviour
        is defined here): a mixed batch of Falcon-512 / Falcon-1024 statements, each ``(logn, sig, pk, hm)``, is
        grouped by parameter set, each group goes through one batched engine call, and the results come back in
        input order as ``[(witness u64[W,4], instance u64[I,4], status)]``."""
        groups = {9: [], 10: []}
        for idx, (logn, sig, pk, hm) in enumerate(items):
            if logn not in groups:
                raise ValueError("logn must be 9 or 10")
            groups[logn].append((idx, sig, pk, hm))
        out = [None] * len(items)
        for logn, members in groups.items():
            if not members:
                continue
            sig, pk, hm = (np.stack([np.asarray(m[k], dtype=np.uint16) for m in members]) for k in (1, 2, 3))
            wit, inst, st = self.witness_ntt_verify(logn, sig, pk, hm, encoding, strict)
            for j, m in enumerate(members):
                out[m[0]] = (wit[j], inst[j], int(st[j]))
        return out

    def ntt_modq(self, logn, poly, encoding=ENC_MONTGOMERY):
        """NTTPolyVar::ntt_circuit alone -> (witness u64[batch, 29N, 4], ntt u16[batch, N], status)."""
        n = 1 << logn
        poly = _u16(poly, n)
        batch = poly.shape[0]
        wit = np.zeros((batch, 29 * n, 4), dtype=np.uint64)
        out = np.zeros((batch, n), dtype=np.uint16)
        st = np.zeros(batch, dtype=np.int32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        check(self._lib.frw_ntt_modq(self._ctx, logn, batch, p(poly), encoding, p(wit), p(out), p(st)), "frw_ntt_modq")
        return wit, out, st

    def prepare_inputs(self, logn, pks, msgs, sigs):
        """(pk bytes, msg bytes, sig bytes) per signature -> (sig, pk, hm) uint16[batch, N] + status, i.e. what the
        reference derives with falcon-rust at falcon_ntt.rs:27-28,44 (decode + SHAKE256 hash-to-point, on the GPU)."""
        batch = len(pks)
        if len(msgs) != batch or len(sigs) != batch:
            raise ValueError("batch mismatch")
        sig_len = len(sigs[0]) if batch else SIG_LEN[logn]
        if any(len(p) != PK_LEN[logn] for p in pks) or any(len(s) != sig_len for s in sigs):
            raise ValueError("public keys must be %d bytes and signatures of one common length" % PK_LEN[logn])
        n = 1 << logn
        pkb = np.frombuffer(b"".join(pks), dtype=np.uint8)
        sgb = np.frombuffer(b"".join(sigs), dtype=np.uint8)
        blob = np.frombuffer(b"".join(msgs) or b"\0", dtype=np.uint8)
        off = np.zeros(batch + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(m) for m in msgs])
        out = [np.zeros((batch, n), dtype=np.uint16) for _ in range(3)]
        st = np.zeros(batch, dtype=np.int32)
        p = lambda x: x.ctypes.data_as(C.c_void_p)
        check(self._lib.frw_prepare_inputs(self._ctx, logn, batch, p(pkb), p(sgb), sig_len, p(blob), p(off),
                                           p(out[0]), p(out[1]), p(out[2]), p(st)), "frw_prepare_inputs")
        return out[0], out[1], out[2], st

    def gadget(self, kind, a, b=None, encoding=ENC_MONTGOMERY):
        """Stand-alone gadget blocks (frw_gadget): a = python ints; returns (blocks u64[count, BLK, 4], status)."""
        blk = self._lib.frw_gadget_block_len(kind)
        if blk < 0:
            raise ValueError("unknown gadget kind %r" % (kind,))
        vals = [int(x) for x in a]
        count = len(vals)
        if kind == G_MOD_Q:
            if any(v >> 160 for v in vals):
                raise ValueError("mod_q input exceeds 160 bits (the ladder never produces more)")
            arr = np.array([[(v >> (32 * i)) & 0xFFFFFFFF for i in range(5)] for v in vals], dtype=np.uint32)
        else:
            arr = np.array(vals, dtype=np.uint64)
        barr = np.array([int(x) for x in b], dtype=np.uint64) if b is not None else None
        out = np.zeros((count, blk, 4), dtype=np.uint64)
        st = np.zeros(count, dtype=np.int32)
        p = lambda x: x.ctypes.data_as(C.c_void_p) if x is not None else None
        check(self._lib.frw_gadget(self._ctx, kind, count, p(arr), p(barr), encoding, p(out), p(st)), "frw_gadget")
        return out, st

    # ---- device buffers (torch tensors or raw pointers) ---------------------------------
    @staticmethod
    def _ptr(t):
        return C.c_void_p(t if isinstance(t, int) else t.data_ptr())

    def witness_ntt_verify_dev(self, logn, batch, d_sig, d_pk, d_hm, d_wit, d_inst, d_status,
                               encoding=ENC_MONTGOMERY, stream=0):
        check(self._lib.frw_witness_ntt_verify_dev(self._ctx, logn, batch, self._ptr(d_sig), self._ptr(d_pk),
                                                   self._ptr(d_hm), encoding, self._ptr(d_wit), self._ptr(d_inst),
                                                   self._ptr(d_status), C.c_void_p(stream)),
              "frw_witness_ntt_verify_dev")

    def witness_ntt_verify_compact_dev(self, logn, batch, d_sig, d_pk, d_hm, d_compact, d_status, stream=0):
        """FRW_ENC_COMPACT producer: d_compact = batch x compact_layout(logn).bytes_per_signature bytes."""
        check(self._lib.frw_witness_ntt_verify_compact_dev(self._ctx, logn, batch, self._ptr(d_sig), self._ptr(d_pk),
                                                           self._ptr(d_hm), self._ptr(d_compact), self._ptr(d_status),
                                                           C.c_void_p(stream)), "frw_witness_ntt_verify_compact_dev")

    def expand_dev(self, logn, batch, d_compact, d_wit, d_inst, stream=0):
        """compact -> the arkworks witness / instance buffers (FRW_ENC_MONTGOMERY bytes)."""
        check(self._lib.frw_expand_dev(self._ctx, logn, batch, self._ptr(d_compact), self._ptr(d_wit), self._ptr(d_inst),
                                       C.c_void_p(stream)), "frw_expand_dev")

    def launch_shape(self, logn, batch, encoding=ENC_MONTGOMERY):
        """{grid, resident workgroups per CU, CUs, split_signatures} of a witness launch of `batch` signatures
        (split_signatures = the ragged tail beyond the last full round of the grid, cut into five work items each)."""
        out = (C.c_int32 * 4)()
        check(self._lib.frw_diag_launch_shape(self._ctx, logn, encoding, batch, C.byref(out)), "frw_diag_launch_shape")
        return {"grid": out[0], "resident_per_cu": out[1], "cus": out[2], "split_signatures": out[3]}

    def witness_dual_ntt_verify_dev(self, logn, batch, d_sig, d_pk, d_hm, d_wit, d_inst, d_status,
                                    encoding=ENC_MONTGOMERY, stream=0):
        check(self._lib.frw_witness_dual_ntt_verify_dev(self._ctx, logn, batch, self._ptr(d_sig), self._ptr(d_pk),
                                                        self._ptr(d_hm), encoding, self._ptr(d_wit), self._ptr(d_inst),
                                                        self._ptr(d_status), C.c_void_p(stream)),
              "frw_witness_dual_ntt_verify_dev")

    def ntt_modq_dev(self, logn, batch, d_poly, d_wit, d_ntt, d_status, encoding=ENC_MONTGOMERY, stream=0):
        check(self._lib.frw_ntt_modq_dev(self._ctx, logn, batch, self._ptr(d_poly), encoding, self._ptr(d_wit),
                                         self._ptr(d_ntt), self._ptr(d_status), C.c_void_p(stream)),
              "frw_ntt_modq_dev")

    def diag_write_stream_dev(self, d_buf, nbytes, slab_bytes, stream=0):
        check(self._lib.frw_diag_write_stream_dev(self._ctx, self._ptr(d_buf), nbytes, slab_bytes, C.c_void_p(stream)),
              "frw_diag_write_stream_dev")

    def r1cs_load(self, circuit, logn):
        """Device-resident A/B/C of circuit 0 (NTT) / 1 (dual NTT) for r1cs_check_dev; free with r1cs_free."""
        h = C.c_void_p()
        check(self._lib.frw_r1cs_load(self.device, circuit, logn, C.byref(h)), "frw_r1cs_load")
        return h

    def r1cs_free(self, handle):
        self._lib.frw_r1cs_free(handle)

    def r1cs_load_aggregate(self, logns):
        """The constraint system of an aggregate statement: FalconNTTVerificationCircuit once per entry of `logns` (9 / 10, in
        order) on one system.  The handle goes wherever an r1cs_load handle goes; its witness / instance vectors are the
        aggregate's own (aggregate_assign_dev makes them).  Free with r1cs_free."""
        arr = np.ascontiguousarray(logns, dtype=np.int32)
        h = C.c_void_p()
        check(self._lib.frw_r1cs_load_aggregate(self.device, len(arr), arr.ctypes.data_as(C.c_void_p), C.byref(h)), "frw_r1cs_load_aggregate")
        return h

    def r1cs_info(self, handle):
        from ._lib import R1csInfoStruct
        q = R1csInfoStruct()
        check(self._lib.frw_r1cs_info(handle, C.byref(q)), "frw_r1cs_info")
        return q

    def aggregate_assign_dev(self, handle, d_wit512, d_inst512, d_wit1024, d_inst1024, d_wit, d_inst, stream=0):
        """instance_assignment / witness_assignment of the aggregate from the per-parameter-set batches of the witness entry points
        (either pair may be None when the aggregate has no such statement)."""
        P = lambda t: self._ptr(t) if t is not None else None
        check(self._lib.frw_aggregate_assign_dev(handle, P(d_wit512), P(d_inst512), P(d_wit1024), P(d_inst1024), self._ptr(d_wit),
                                                 self._ptr(d_inst), C.c_void_p(stream)), "frw_aggregate_assign_dev")

    def groth16_setup_r1cs(self, handle, alpha, beta, gamma, delta, t, mode=KEY_AUTO, rank=0, world=1, want_vk=True):
        """groth16_setup for the system behind an r1cs handle (a per-signature circuit or an aggregate statement).
        mode: KEY_TABLES (window tables), KEY_BARE (the points only, made on the device end to end), KEY_AUTO by size;
        rank / world: one slice of a key in slices (bare)."""
        from ._lib import Groth16KeyOpts
        tox = np.frombuffer(b"".join(int(x).to_bytes(32, "little") for x in (alpha, beta, gamma, delta, t)), dtype=np.uint64).copy()
        ni = int(self.r1cs_info(handle).num_instance)
        vk = np.zeros(84 + 12 * ni, dtype=np.uint64)
        h = C.c_void_p()
        opts = Groth16KeyOpts(int(mode), int(rank), int(world))
        check(self._lib.frw_groth16_setup_r1cs_opts(handle, tox.ctypes.data_as(C.c_void_p), C.byref(opts), C.byref(h),
                                                    vk.ctypes.data_as(C.c_void_p) if want_vk else None), "frw_groth16_setup_r1cs")
        if not want_vk:
            return h, None
        return h, {"alpha_g1": vk[:12], "beta_g2": vk[12:36], "gamma_g2": vk[36:60], "delta_g2": vk[60:84],
                   "gamma_abc_g1": vk[84:].reshape(-1, 12)}

    def r1cs_check_dev(self, handle, batch, d_wit, d_inst, d_num_unsatisfied, stream=0):
        check(self._lib.frw_r1cs_check_dev(handle, batch, self._ptr(d_wit), self._ptr(d_inst),
                                           self._ptr(d_num_unsatisfied), C.c_void_p(stream)), "frw_r1cs_check_dev")

    def r1cs_eval_dev(self, handle, batch, d_wit, d_inst, d_num_unsatisfied, d_abc, stream=0):
        """r1cs_check_dev + A z, B z, C z into d_abc (int64[batch, 3, C, 4], Montgomery)."""
        check(self._lib.frw_r1cs_eval_dev(handle, batch, self._ptr(d_wit), self._ptr(d_inst),
                                          self._ptr(d_num_unsatisfied), self._ptr(d_abc), C.c_void_p(stream)),
              "frw_r1cs_eval_dev")

    def r1cs_eval_scratch_bytes(self, handle, batch, with_products):
        return int(self._lib.frw_r1cs_eval_scratch_bytes(handle, batch, 1 if with_products else 0))

    def r1cs_eval_scratch_dev(self, handle, batch, d_wit, d_inst, d_num_unsatisfied, d_abc, d_scratch, scratch_bytes, stream=0):
        """frw_r1cs_eval_dev / frw_r1cs_check_dev (d_abc=None) with the caller's scratch: allocates nothing, capture-safe."""
        check(self._lib.frw_r1cs_eval_scratch_dev(handle, batch, self._ptr(d_wit), self._ptr(d_inst), self._ptr(d_num_unsatisfied),
                                                  self._ptr(d_abc) if d_abc is not None else None,
                                                  self._ptr(d_scratch) if d_scratch is not None else None, scratch_bytes,
                                                  C.c_void_p(stream)), "frw_r1cs_eval_scratch_dev")

    # ---- multi-scalar multiplication over BLS12-381 G1 (frw_msm.hip) --------------------------------------------------
    def msm_g1_load(self, bases, narrow=False, bare=False, wide=False):
        """bases: uint64[n, 12] (ark-ff's bytes of n affine points, zeros = infinity) -> handle; free with msm_free.
        narrow: 8-bit windows (128 buckets) instead of 16-bit ones: for scalars that are mostly zero, one or small.
        bare: the points only, no window table (the sums run window by window); wide: a dense bare handle on thirteen 20-bit windows
        whatever its size (what handles of 2^23 points and more run on anyway)."""
        bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, 12)
        h = C.c_void_p()
        if bare:
            check(self._lib.frw_msm_g1_load_bare(self.device, bases.shape[0], bases.ctypes.data_as(C.c_void_p), 2 if wide else 1 if narrow else 0, C.byref(h)),
                  "frw_msm_g1_load_bare")
            return h
        fn = self._lib.frw_msm_g1_load_narrow if narrow else self._lib.frw_msm_g1_load
        check(fn(self.device, bases.shape[0], bases.ctypes.data_as(C.c_void_p), C.byref(h)), "frw_msm_g1_load")
        return h

    def g1_fixed_base(self, scalars):
        """k_i G1 on the device: canonical scalars uint64[count, 4] -> uint64[count, 12] (ark-ff's bytes)."""
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros((scalars.shape[0], 12), dtype=np.uint64)
        check(self._lib.frw_g1_fixed_base(self.device, scalars.shape[0], scalars.ctypes.data_as(C.c_void_p),
                                          out.ctypes.data_as(C.c_void_p)), "frw_g1_fixed_base")
        return out

    def g2_fixed_base(self, scalars):
        """k_i G2 on the device: canonical scalars uint64[count, 4] -> uint64[count, 24] (ark-ff's bytes)."""
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros((scalars.shape[0], 24), dtype=np.uint64)
        check(self._lib.frw_g2_fixed_base(self.device, scalars.shape[0], scalars.ctypes.data_as(C.c_void_p),
                                          out.ctypes.data_as(C.c_void_p)), "frw_g2_fixed_base")
        return out

    def msm_g2_load(self, bases, narrow=False, bare=False, wide=False):
        bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, 24)
        h = C.c_void_p()
        if bare:
            check(self._lib.frw_msm_g2_load_bare(self.device, bases.shape[0], bases.ctypes.data_as(C.c_void_p), 2 if wide else 1 if narrow else 0, C.byref(h)),
                  "frw_msm_g2_load_bare")
            return h
        fn = self._lib.frw_msm_g2_load_narrow if narrow else self._lib.frw_msm_g2_load
        check(fn(self.device, bases.shape[0], bases.ctypes.data_as(C.c_void_p), C.byref(h)), "frw_msm_g2_load")
        return h

    def msm_g2_dev(self, handle, batch, d_scalars, scalar_stride, montgomery, d_out, d_workspace, workspace_bytes, stream=0):
        check(self._lib.frw_msm_g2_dev(handle, batch, self._ptr(d_scalars), scalar_stride, 1 if montgomery else 0, self._ptr(d_out),
                                       self._ptr(d_workspace), workspace_bytes, C.c_void_p(stream)), "frw_msm_g2_dev")

    def msm_free(self, handle):
        self._lib.frw_msm_free(handle)

    def msm_info(self, handle):
        from ._lib import MsmInfoStruct
        info = MsmInfoStruct()
        check(self._lib.frw_msm_info(handle, C.byref(info)), "frw_msm_info")
        return info

    def msm_g1_dev(self, handle, batch, d_scalars, scalar_stride, montgomery, d_out, d_workspace, workspace_bytes, stream=0):
        check(self._lib.frw_msm_g1_dev(handle, batch, self._ptr(d_scalars), scalar_stride, 1 if montgomery else 0, self._ptr(d_out),
                                       self._ptr(d_workspace), workspace_bytes, C.c_void_p(stream)), "frw_msm_g1_dev")

    def groth16_msm_h_dev(self, handle, batch, d_h, domain_size, d_out, d_workspace, workspace_bytes, stream=0):
        check(self._lib.frw_groth16_msm_h_dev(handle, batch, self._ptr(d_h), domain_size, self._ptr(d_out), self._ptr(d_workspace),
                                              workspace_bytes, C.c_void_p(stream)), "frw_groth16_msm_h_dev")

    # ---- a whole Groth16 proof per signature ---------------------------------------------------------------------------
    def groth16_pk_load(self, num_instance, num_witness, domain_size, alpha_g1, beta_g1, delta_g1, beta_g2, delta_g2, a_query,
                        b_g1_query, b_g2_query, h_query, l_query, mode=KEY_TABLES, rank=0, world=1):
        """The proving key's elements as uint64 arrays in ark-ff's bytes (G1 rows of 12, G2 rows of 24) -> handle.
        mode / rank / world: as groth16_setup_r1cs (every rank passes the WHOLE key and keeps its slice)."""
        from ._lib import Groth16PkDesc, Groth16KeyOpts
        arrs = [np.ascontiguousarray(a, dtype=np.uint64) for a in (alpha_g1, beta_g1, delta_g1, beta_g2, delta_g2, a_query, b_g1_query,
                                                                   b_g2_query, h_query, l_query)]
        nv = num_instance + num_witness
        want = [12, 12, 12, 24, 24, nv * 12, nv * 12, nv * 24, (domain_size - 1) * 12, num_witness * 12]
        if [a.size for a in arrs] != want:
            raise ValueError("proving key: wrong array sizes")
        d = Groth16PkDesc(num_instance, num_witness, domain_size, *[a.ctypes.data_as(C.c_void_p) for a in arrs])
        h = C.c_void_p()
        opts = Groth16KeyOpts(int(mode), int(rank), int(world))
        check(self._lib.frw_groth16_pk_load_opts(self.device, C.byref(d), C.byref(opts), C.byref(h)), "frw_groth16_pk_load")
        return h

    def groth16_pk_info(self, pk):
        from ._lib import Groth16PkInfoStruct
        info = Groth16PkInfoStruct()
        check(self._lib.frw_groth16_pk_info(pk, C.byref(info)), "frw_groth16_pk_info")
        return info

    def groth16_pk_query(self, pk, which):
        """One of the key's five tables (0: h_query, 1: a, 2: b_g1, 3: l, 4: b_g2) as a BORROWED msm handle (never msm_free it)."""
        h = self._lib.frw_groth16_pk_query(pk, int(which))
        if not h:
            raise FrwError(-1, "frw_groth16_pk_query", "no such query")
        return C.c_void_p(h)

    def groth16_prove_partial_dev(self, pk, r1cs, batch, d_wit, d_inst, rs, d_partial, d_workspace, workspace_bytes, d_num_unsatisfied=None, stream=0):
        """One rank's partial sums of a key in slices: d_partial int64[batch, 72] = A | B1' | L | H | B (ark-ff's affine bytes)."""
        rs = np.ascontiguousarray(rs, dtype=np.uint64).reshape(batch, 2, 4)
        check(self._lib.frw_groth16_prove_partial_dev(pk, r1cs, batch, self._ptr(d_wit), self._ptr(d_inst), rs.ctypes.data_as(C.c_void_p),
                                                      self._ptr(d_partial), self._ptr(d_num_unsatisfied) if d_num_unsatisfied is not None else None,
                                                      self._ptr(d_workspace), workspace_bytes, C.c_void_p(stream)), "frw_groth16_prove_partial_dev")

    def groth16_prove_combine_dev(self, pk, world, d_partials, rs, d_proof, d_workspace, workspace_bytes, stream=0):
        """All ranks' partial sums (int64[world, 72], rank order) -> the proof int64[48]."""
        rs = np.ascontiguousarray(rs, dtype=np.uint64).reshape(2, 4)
        check(self._lib.frw_groth16_prove_combine_dev(pk, world, self._ptr(d_partials), rs.ctypes.data_as(C.c_void_p), self._ptr(d_proof),
                                                      self._ptr(d_workspace), workspace_bytes, C.c_void_p(stream)), "frw_groth16_prove_combine_dev")

    def diag_groth16_side_counts(self, pk, d_z, d_workspace, workspace_bytes, stream=0):
        """(rows of b_g1_query / b_g2_query that hold a point, digits and ones over all rows, digits and ones over those rows): the point
        additions of the witness-side sums of a key of bare handles for the scalars d_z (z ++ [1, r, s] of the key's slice)."""
        out = np.zeros(5, dtype=np.uint64)
        check(self._lib.frw_diag_groth16_side_counts(pk, self._ptr(d_z), self._ptr(d_workspace), workspace_bytes, C.c_void_p(stream),
                                                     out.ctypes.data_as(C.c_void_p)), "frw_diag_groth16_side_counts")
        return [int(v) for v in out]

    def diag_poly_eval_dev(self, d_coeffs, n, t):
        """p(t) for the polynomial whose n coefficients (ark-ff's Montgomery form) are in device memory; t, result: Python integers."""
        tt = np.frombuffer(int(t).to_bytes(32, "little"), dtype=np.uint64).copy()
        out = np.zeros(4, dtype=np.uint64)
        check(self._lib.frw_diag_poly_eval_dev(self.device, n, self._ptr(d_coeffs), tt.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)),
              "frw_diag_poly_eval_dev")
        return int.from_bytes(out.tobytes(), "little")

    def groth16_setup(self, circuit, logn, alpha, beta, gamma, delta, t):
        """generate_parameters with the given toxic waste (Python integers) -> (proving-key handle, verifying key dict of uint64 arrays)."""
        tox = np.frombuffer(b"".join(int(x).to_bytes(32, "little") for x in (alpha, beta, gamma, delta, t)), dtype=np.uint64).copy()
        L = layout_dual(logn) if circuit else layout(logn)
        vk = np.zeros(84 + 12 * L.num_instance, dtype=np.uint64)
        h = C.c_void_p()
        check(self._lib.frw_groth16_setup(self.device, circuit, logn, tox.ctypes.data_as(C.c_void_p), C.byref(h),
                                          vk.ctypes.data_as(C.c_void_p)), "frw_groth16_setup")
        return h, {"alpha_g1": vk[:12], "beta_g2": vk[12:36], "gamma_g2": vk[36:60], "delta_g2": vk[60:84],
                   "gamma_abc_g1": vk[84:].reshape(-1, 12)}

    def groth16_pk_free(self, handle):
        self._lib.frw_groth16_pk_free(handle)

    def groth16_workspace_bytes(self, pk, r1cs, in_flight):
        return int(self._lib.frw_groth16_workspace_bytes(pk, r1cs, in_flight))

    def groth16_prove_dev(self, pk, r1cs, batch, d_wit, d_inst, rs, d_proofs, d_workspace, workspace_bytes, d_num_unsatisfied=None, stream=0):
        """rs: host uint64[batch, 2, 4] (r, s per proof, canonical); d_proofs: int64[batch, 48] = A | B | C in ark-ff's bytes."""
        rs = np.ascontiguousarray(rs, dtype=np.uint64).reshape(batch, 2, 4)
        check(self._lib.frw_groth16_prove_dev(pk, r1cs, batch, self._ptr(d_wit), self._ptr(d_inst), rs.ctypes.data_as(C.c_void_p),
                                              self._ptr(d_proofs), self._ptr(d_num_unsatisfied) if d_num_unsatisfied is not None else None,
                                              self._ptr(d_workspace), workspace_bytes, C.c_void_p(stream)), "frw_groth16_prove_dev")

    def groth16_prove_rs_dev(self, pk, r1cs, batch, d_wit, d_inst, d_rs, d_proofs, d_workspace, workspace_bytes, d_num_unsatisfied=None, stream=0):
        """The same with the blinding factors in device memory (d_rs: int64[batch, 2, 4]): stream-ordered throughout, capturable."""
        check(self._lib.frw_groth16_prove_rs_dev(pk, r1cs, batch, self._ptr(d_wit), self._ptr(d_inst), self._ptr(d_rs),
                                                 self._ptr(d_proofs), self._ptr(d_num_unsatisfied) if d_num_unsatisfied is not None else None,
                                                 self._ptr(d_workspace), workspace_bytes, C.c_void_p(stream)), "frw_groth16_prove_rs_dev")

    def qap_info(self, handle):
        """Domain of the QAP witness map for the loaded matrices: (log n, n, C, I, workspace bytes per signature)."""
        from ._lib import QapInfoStruct
        q = QapInfoStruct()
        check(self._lib.frw_qap_info(handle, C.byref(q)), "frw_qap_info")
        return q

    def qap_witness_map_dev(self, handle, batch, d_wit, d_inst, d_h, d_workspace, workspace_bytes, d_num_unsatisfied=None,
                            stream=0):
        """ark-groth16's R1CStoQAP::witness_map for every signature of a resident batch: d_h = int64[batch, n, 4]
        (Montgomery), coefficient k of h(X) = (A B - C)(X) / (X^n - 1) at index k."""
        check(self._lib.frw_qap_witness_map_dev(handle, batch, self._ptr(d_wit), self._ptr(d_inst), self._ptr(d_h),
                                                self._ptr(d_num_unsatisfied) if d_num_unsatisfied is not None else None,
                                                self._ptr(d_workspace), workspace_bytes, C.c_void_p(stream)),
              "frw_qap_witness_map_dev")

    def qap_quotient_dev(self, handle, batch, d_wit, d_inst, d_h, d_workspace, workspace_bytes, d_num_unsatisfied=None, stream=0):
        """h with six transforms instead of seven: equal to qap_witness_map_dev's wherever d_num_unsatisfied is 0."""
        check(self._lib.frw_qap_quotient_dev(handle, batch, self._ptr(d_wit), self._ptr(d_inst), self._ptr(d_h),
                                             self._ptr(d_num_unsatisfied) if d_num_unsatisfied is not None else None,
                                             self._ptr(d_workspace), workspace_bytes, C.c_void_p(stream)),
              "frw_qap_quotient_dev")

    def qap_witness_map(self, handle, witness, instance):
        """Host arrays (uint64[batch, W, 4], uint64[batch, I, 4], Montgomery, as witness_ntt_verify returns them) ->
        (h uint64[batch, n, 4], unsatisfied rows uint32[batch])."""
        witness = np.ascontiguousarray(witness, dtype=np.uint64)
        instance = np.ascontiguousarray(instance, dtype=np.uint64)
        batch = witness.shape[0]
        n = int(self.qap_info(handle).domain_size)
        h = np.empty((batch, n, 4), dtype=np.uint64)
        bad = np.empty(batch, dtype=np.uint32)
        check(self._lib.frw_qap_witness_map(handle, batch, witness.ctypes.data_as(C.c_void_p), instance.ctypes.data_as(C.c_void_p),
                                            h.ctypes.data_as(C.c_void_p), bad.ctypes.data_as(C.c_void_p)), "frw_qap_witness_map")
        return h, bad

    def digest_dev(self, d_buf, words_per_item, items, d_out, stream=0):
        check(self._lib.frw_digest_dev(self._ctx, self._ptr(d_buf), words_per_item, items, self._ptr(d_out),
                                       C.c_void_p(stream)), "frw_digest_dev")
