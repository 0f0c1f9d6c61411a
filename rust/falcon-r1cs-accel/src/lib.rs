//! `FalconNTTVerificationCircuit` with its witness assignment computed on the GPU.
//!
//! The reference circuit (`falcon-r1cs/src/circuits/falcon_ntt.rs:8-123`) is used as it is: in prove mode
//! [`AcceleratedNTTCircuit`] runs it with the constraint system switched to `SynthesisMode::Setup`, so that arkworks
//! allocates every variable and records every constraint but evaluates no value closure (the gadgets substitute
//! `F::one()` themselves in that mode: `arithmetics.rs:121-125`, `range_proofs.rs:49-53`, `:106-110`, `:197-201`,
//! `:295-299`), then switches back and installs the two assignment vectors from the engine.  The engine's buffers
//! are in arkworks' allocation order and in `Fp256`'s in-memory (Montgomery) representation, so installing them is a
//! reinterpretation, not a conversion.
//!
//! Not compiled in the repository this file ships in (no Rust toolchain there); see `rust/README.md`.

use std::os::raw::c_int;

use ark_bls12_381::Fr;
use ark_ff::{BigInteger256, Fp256};
use ark_relations::r1cs::{ConstraintSynthesizer, ConstraintSystemRef, Result as R1csResult, SynthesisError, SynthesisMode};
use falcon_r1cs::FalconNTTVerificationCircuit;
use falcon_rust::{NTTPolynomial, Polynomial, PublicKey, Signature, LOG_N, N};
use frw_sys as sys;

/// A failed engine call: the negative `FRW_E_*` code and its text.
#[derive(Debug)]
pub struct EngineError {
    pub code: i32,
    pub what: String,
}

fn check(code: c_int) -> Result<(), EngineError> {
    if code == sys::FRW_OK {
        return Ok(());
    }
    let what = unsafe { std::ffi::CStr::from_ptr(sys::frw_strerror(code)).to_string_lossy().into_owned() };
    Err(EngineError { code, what })
}

/// One context on one HIP device.  Like a `ConstraintSystemRef` it is used by one thread at a time.
pub struct Engine {
    ctx: *mut sys::frw_ctx,
}

impl Engine {
    pub fn new(device: i32) -> Result<Self, EngineError> {
        let mut ctx = std::ptr::null_mut();
        check(unsafe { sys::frw_ctx_create(device, &mut ctx) })?;
        Ok(Self { ctx })
    }
}

impl Drop for Engine {
    fn drop(&mut self) {
        unsafe { sys::frw_ctx_destroy(self.ctx) }
    }
}

/// The two assignment vectors of one signature's constraint system, exactly as arkworks would hold them.
#[derive(Clone, Debug)]
pub struct Assignment {
    /// `cs.witness_assignment`: W = 153 N + {50 | 52} elements, allocation order of falcon_ntt.rs:58-122
    pub witness: Vec<Fr>,
    /// `cs.instance_assignment`: [1, pk_ntt, hm_ntt] (falcon_ntt.rs:63,67)
    pub instance: Vec<Fr>,
    /// FRW_ST_OK | FRW_ST_COEFF_RANGE | FRW_ST_NORM_BOUND
    pub status: i32,
}

#[inline]
fn fr_from_montgomery_limbs(l: &[u64]) -> Fr {
    // ark-ff 0.3: `pub struct Fp256<P>(pub BigInteger256, pub PhantomData<P>)`; `Fp256::new` takes the limbs verbatim
    Fp256::new(BigInteger256([l[0], l[1], l[2], l[3]]))
}

impl Engine {
    /// Witness and instance assignment of a batch of statements in ONE engine call.
    /// `strict` mirrors the reference's non-test build: a signature whose norm reaches the bound is an error there
    /// (`range_proofs.rs:114-117,205-208` panic), `FRW_E_RANGE` here.
    pub fn witness_ntt_verify(
        &self,
        statements: &[(&PublicKey, &[u8], &Signature)],
        strict: bool,
    ) -> Result<Vec<Assignment>, EngineError> {
        let batch = statements.len();
        let (mut sig, mut pk, mut hm) = (Vec::with_capacity(batch * N), Vec::with_capacity(batch * N), Vec::with_capacity(batch * N));
        for (p, msg, s) in statements {
            // the same three coefficient vectors the reference derives at falcon_ntt.rs:27-28,44
            let sig_poly: Polynomial = (*s).into();
            let pk_poly: Polynomial = (*p).into();
            let hm_poly = Polynomial::from_hash_of_message(msg, s.nonce());
            sig.extend_from_slice(sig_poly.coeff());
            pk.extend_from_slice(pk_poly.coeff());
            hm.extend_from_slice(hm_poly.coeff());
        }
        let mut layout = sys::frw_layout_t::default();
        check(unsafe { sys::frw_layout(LOG_N as c_int, &mut layout) })?;
        let (w, i) = (layout.num_witness as usize, layout.num_instance as usize);
        let mut wit = vec![0u64; batch * w * 4];
        let mut inst = vec![0u64; batch * i * 4];
        let mut status = vec![0i32; batch];
        check(unsafe {
            sys::frw_witness_ntt_verify(self.ctx, LOG_N as c_int, batch, sig.as_ptr(), pk.as_ptr(), hm.as_ptr(),
                                        sys::FRW_ENC_MONTGOMERY, wit.as_mut_ptr(), inst.as_mut_ptr(),
                                        status.as_mut_ptr(), strict as c_int)
        })?;
        Ok((0..batch)
            .map(|k| Assignment {
                witness: wit[k * w * 4..(k + 1) * w * 4].chunks_exact(4).map(fr_from_montgomery_limbs).collect(),
                instance: inst[k * i * 4..(k + 1) * i * 4].chunks_exact(4).map(fr_from_montgomery_limbs).collect(),
                status: status[k],
            })
            .collect())
    }
}

/// The reference circuit plus the assignment the engine computed for it.
pub struct AcceleratedNTTCircuit {
    inner: FalconNTTVerificationCircuit,
    assignment: Assignment,
}

impl AcceleratedNTTCircuit {
    /// `assignment` must come from [`Engine::witness_ntt_verify`] for the same `(pk, msg, sig)`.
    pub fn build_circuit(pk: PublicKey, msg: Vec<u8>, sig: Signature, assignment: Assignment) -> Self {
        Self { inner: FalconNTTVerificationCircuit::build_circuit(pk, msg, sig), assignment }
    }
}

impl ConstraintSynthesizer<Fr> for AcceleratedNTTCircuit {
    fn generate_constraints(self, cs: ConstraintSystemRef<Fr>) -> R1csResult<()> {
        if cs.is_in_setup_mode() {
            // circuit_specific_setup: structure only, nothing to accelerate (examples/pok_sig.rs:30-31)
            return self.inner.generate_constraints(cs);
        }
        // prove mode: structure pass without value closures ...
        let construct_matrices = cs.should_construct_matrices();
        cs.set_mode(SynthesisMode::Setup);
        let structure = self.inner.generate_constraints(cs.clone());
        cs.set_mode(SynthesisMode::Prove { construct_matrices });
        structure?;
        // ... then the values, wholesale
        let mut sys = cs.borrow_mut().ok_or(SynthesisError::MissingCS)?;
        if sys.num_witness_variables != self.assignment.witness.len()
            || sys.num_instance_variables != self.assignment.instance.len()
        {
            return Err(SynthesisError::Unsatisfiable); // the engine was asked for another parameter set
        }
        sys.witness_assignment = self.assignment.witness;
        sys.instance_assignment = self.assignment.instance;
        Ok(())
    }
}

/// The constraint matrices of one circuit on the device plus the tables of its evaluation domain: what
/// `frw_qap_witness_map` needs.  Built once per (circuit, parameter set), like a proving key.
pub struct R1cs {
    raw: *mut sys::frw_r1cs,
    info: sys::frw_qap_info_t,
}

impl R1cs {
    /// `circuit`: `sys::FRW_CIRCUIT_NTT` or `sys::FRW_CIRCUIT_DUAL_NTT`.
    pub fn load(device: i32, circuit: i32) -> Result<Self, EngineError> {
        let mut raw = std::ptr::null_mut();
        check(unsafe { sys::frw_r1cs_load(device, circuit, LOG_N as c_int, &mut raw) })?;
        let mut info = sys::frw_qap_info_t::default();
        check(unsafe { sys::frw_qap_info(raw, &mut info) })?;
        Ok(Self { raw, info })
    }

    /// Size of ark-poly's `Radix2EvaluationDomain::new(num_constraints + num_instance_variables)`.
    pub fn domain_size(&self) -> usize {
        self.info.domain_size as usize
    }

    /// ark-groth16's `R1CStoQAP::witness_map` for each `(instance_assignment, witness_assignment)` pair: the coefficients
    /// of h(X) = (A(X) B(X) - C(X)) / (X^n - 1), `domain_size()` of them, computed on the GPU.
    pub fn witness_map(&self, full: &[(&[Fr], &[Fr])]) -> Result<Vec<Vec<Fr>>, EngineError> {
        let (ni, n) = (self.info.num_instance as usize, self.domain_size());
        let mut wit: Vec<u64> = Vec::new();
        let mut inst: Vec<u64> = Vec::new();
        for (i, w) in full {
            assert_eq!(i.len(), ni, "instance_assignment with the constant one first");
            inst.extend(i.iter().flat_map(|e| (e.0).0));          // Fp256(BigInteger256([u64; 4]), _): Montgomery limbs
            wit.extend(w.iter().flat_map(|e| (e.0).0));
        }
        let mut h = vec![0u64; full.len() * n * 4];
        check(unsafe {
            sys::frw_qap_witness_map(self.raw, full.len(), wit.as_ptr(), inst.as_ptr(), h.as_mut_ptr(), std::ptr::null_mut())
        })?;
        Ok(h.chunks_exact(n * 4)
            .map(|one| one.chunks_exact(4).map(|l| Fp256::new(BigInteger256([l[0], l[1], l[2], l[3]]))).collect())
            .collect())
    }
}

impl Drop for R1cs {
    fn drop(&mut self) {
        unsafe { sys::frw_r1cs_free(self.raw) }
    }
}

/// ONE constraint system for several signatures -- what `falcon-aggregate-sig` (upstream a stub: `falcon-aggregate-sig/src/main.rs:1-3`)
/// would build by running `FalconNTTVerificationCircuit::generate_constraints` once per `(pk, msg, sig)` on the same
/// `ConstraintSystemRef` -- as the engine holds it: the per-signature systems' blocks, never the aggregate's own matrices.
/// (`LOG_N` is a cargo feature of falcon-rust, so one build of this crate aggregates one parameter set; the C ABI mixes them freely.)
pub struct AggregateR1cs {
    inner: R1cs,
    statements: usize,
}

impl AggregateR1cs {
    pub fn load(device: i32, statements: usize) -> Result<Self, EngineError> {
        let logn = vec![LOG_N as i32; statements];
        let mut raw = std::ptr::null_mut();
        check(unsafe { sys::frw_r1cs_load_aggregate(device, statements, logn.as_ptr(), &mut raw) })?;
        let mut info = sys::frw_qap_info_t::default();
        check(unsafe { sys::frw_qap_info(raw, &mut info) })?;
        Ok(Self { inner: R1cs { raw, info }, statements })
    }

    /// `instance_assignment` / `witness_assignment` of the aggregate's constraint system from the statements' own: the
    /// constant one, then every statement's public inputs; the statements' witnesses end to end.  (Device-resident batches go
    /// through `frw_aggregate_assign_dev` instead.)
    pub fn assign(&self, parts: &[Assignment]) -> Assignment {
        assert_eq!(parts.len(), self.statements);
        let mut instance = vec![parts[0].instance[0]];
        let mut witness = Vec::new();
        for p in parts {
            instance.extend_from_slice(&p.instance[1..]);
            witness.extend_from_slice(&p.witness);
        }
        Assignment { witness, instance }
    }

    /// The aggregate's h(X): `R1CStoQAP::witness_map` over `Radix2EvaluationDomain::new(sum C_i + 1 + sum 2 N_i)`.
    pub fn witness_map(&self, full: &Assignment) -> Result<Vec<Fr>, EngineError> {
        Ok(self.inner.witness_map(&[(&full.instance[..], &full.witness[..])])?.remove(0))
    }

    pub fn domain_size(&self) -> usize {
        self.inner.domain_size()
    }
}

/// The public inputs a verifier feeds `verify_proof` with, as examples/pok_sig.rs:33-45 computes them.
pub fn public_inputs(pk: &PublicKey, msg: &[u8], sig: &Signature) -> Vec<Fr> {
    let pk_ntt = NTTPolynomial::from(&Polynomial::from(pk));
    let hm_ntt = NTTPolynomial::from(&Polynomial::from_hash_of_message(msg, sig.nonce()));
    pk_ntt.coeff().iter().chain(hm_ntt.coeff().iter()).map(|e| Fr::from(*e)).collect()
}

/// ... and for an aggregate statement: the statements' public inputs one after the other.
pub fn aggregate_public_inputs(statements: &[(PublicKey, Vec<u8>, Signature)]) -> Vec<Fr> {
    statements.iter().flat_map(|(pk, msg, sig)| public_inputs(pk, msg, sig)).collect()
}

#[cfg(test)]
mod tests {
    use super::*;
    use ark_relations::r1cs::ConstraintSystem;
    use falcon_rust::KeyPair;

    /// The reference's own end-to-end test (falcon_ntt.rs:133-160) on the accelerated circuit, plus the one check that
    /// would PIN the engine's parity: its witness equals, element by element, the one arkworks computes on the CPU.
    #[test]
    fn accelerated_circuit_is_satisfied_and_equals_the_cpu_witness() {
        let keypair = KeyPair::keygen();
        let msg = "testing message";
        let sig = keypair.secret_key.sign_with_seed("test seed".as_ref(), msg.as_ref());
        assert!(keypair.public_key.verify(msg.as_ref(), &sig));

        let engine = Engine::new(0).expect("an MI355X and libfrw.so");
        let a = engine.witness_ntt_verify(&[(&keypair.public_key, msg.as_bytes(), &sig)], true).unwrap().remove(0);
        assert_eq!(a.status, frw_sys::FRW_ST_OK);

        // CPU: the reference as it is
        let cpu = ConstraintSystem::<Fr>::new_ref();
        FalconNTTVerificationCircuit::build_circuit(keypair.public_key, msg.as_bytes().to_vec(), sig)
            .generate_constraints(cpu.clone())
            .unwrap();
        assert!(cpu.is_satisfied().unwrap());

        // GPU values, same structure
        let gpu = ConstraintSystem::<Fr>::new_ref();
        AcceleratedNTTCircuit::build_circuit(keypair.public_key, msg.as_bytes().to_vec(), sig, a)
            .generate_constraints(gpu.clone())
            .unwrap();
        assert!(gpu.is_satisfied().unwrap());
        assert_eq!(gpu.num_constraints(), cpu.num_constraints());
        let (g, c) = (gpu.borrow().unwrap(), cpu.borrow().unwrap());
        assert_eq!(g.instance_assignment, c.instance_assignment);
        assert_eq!(g.witness_assignment, c.witness_assignment);
    }

    /// What would pin the QAP witness map: h from the GPU satisfies, at a random point, the identity that defines
    /// ark-groth16's `R1CStoQAP::witness_map` output -- A(tau) B(tau) - C(tau) = h(tau) Z(tau) with the Lagrange
    /// coefficients, the matrices and the assignment all taken from arkworks itself (ark-poly 0.3 `EvaluationDomain`,
    /// `cs.to_matrices()` after `finalize()`), none from this repository.
    #[test]
    fn gpu_witness_map_satisfies_arkworks_qap_identity() {
        use ark_ff::{One, UniformRand, Zero};
        use ark_poly::{EvaluationDomain, GeneralEvaluationDomain};
        let keypair = KeyPair::keygen();
        let msg = "testing message";
        let sig = keypair.secret_key.sign_with_seed("test seed".as_ref(), msg.as_ref());
        let cs = ConstraintSystem::<Fr>::new_ref();
        FalconNTTVerificationCircuit::build_circuit(keypair.public_key, msg.as_bytes().to_vec(), sig)
            .generate_constraints(cs.clone())
            .unwrap();
        cs.finalize();
        let m = cs.to_matrices().unwrap();
        let (inst, wit) = {
            let c = cs.borrow().unwrap();
            (c.instance_assignment.clone(), c.witness_assignment.clone())
        };
        let r1cs = R1cs::load(0, frw_sys::FRW_CIRCUIT_NTT).expect("an MI355X and libfrw.so");
        let h = r1cs.witness_map(&[(&inst, &wit)]).unwrap().remove(0);

        let domain = GeneralEvaluationDomain::<Fr>::new(cs.num_constraints() + cs.num_instance_variables()).unwrap();
        assert_eq!(domain.size(), h.len());
        let tau = Fr::rand(&mut ark_std::test_rng());
        let lag = domain.evaluate_all_lagrange_coefficients(tau);
        let z: Vec<Fr> = inst.iter().chain(wit.iter()).cloned().collect();
        let dot = |rows: &Vec<Vec<(Fr, usize)>>, extra: &[Fr]| -> Fr {
            let mut acc = Fr::zero();
            for (i, row) in rows.iter().enumerate() {
                acc += lag[i] * row.iter().map(|(c, j)| *c * z[*j]).sum::<Fr>();
            }
            for (j, e) in extra.iter().enumerate() {
                acc += lag[rows.len() + j] * e;
            }
            acc
        };
        let (a, b, c) = (dot(&m.a, &inst), dot(&m.b, &[]), dot(&m.c, &[]));
        let mut h_tau = Fr::zero();
        let mut p = Fr::one();
        for coeff in &h {
            h_tau += *coeff * p;
            p *= tau;
        }
        assert_eq!(a * b - c, h_tau * domain.evaluate_vanishing_polynomial(tau));
    }
}
