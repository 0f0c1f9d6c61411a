"""ORACLE (test infrastructure only) -- minimal simulation of the arkworks 0.3 R1CS front end.

This file is part of ``oracle/``: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.  It is never on the product path.

PARITY UNPINNED: the reference (``/root/reference``, Rust on arkworks 0.3 + falcon-rust)
cannot be compiled or run in this environment (no cargo/rustc, dependencies un-vendored,
no network) and holds no golden witness vectors.  What *is* pinned is listed in
``oracle/README.md``: the README's variable/constraint count table, every known-answer
test of the reference's gadget unit tests, and the NTT == direct-evaluation property.

What is simulated, and from where
---------------------------------
The reference's gadgets are written against three third-party crates that are NOT under
``/root/reference`` (``falcon-r1cs/Cargo.toml:14-19``):

* ``ark-relations 0.3.0``  -- ``ConstraintSystem`` (``new_input_variable``,
  ``new_witness_variable``, ``new_lc``, ``enforce_constraint``, ``is_satisfied``, the
  ``num_*`` counters, and the two assignment vectors ``instance_assignment`` /
  ``witness_assignment`` with ``instance_assignment[0] == 1``).
* ``ark-r1cs-std 0.3.1``   -- ``FpVar`` (``Constant`` | ``Var(AllocatedFp)``),
  ``Boolean`` (``Is`` | ``Not`` | ``Constant``) over ``AllocatedBool``, and their operator
  semantics: which operations allocate a witness, which only build a symbolic linear
  combination, and in which order.
* ``ark-ff 0.3.0``         -- ``Fp256`` elements; only the value semantics (integers mod p)
  matter here, the Montgomery encoding is applied by ``falcon_gadgets.encode_*``.

Their published behaviour is restated below; each method names the arkworks item it
restates.  The witness *order* of the reference falls out of running the restated gadgets
(``oracle/falcon_gadgets.py``) against this simulator in the reference's own call order,
rather than from a hand-written layout table.
"""
from __future__ import annotations

# BLS12-381 scalar field (ark_ed_on_bls12_381::fq::Fq == ark_bls12_381::Fr), the field every
# test/example of the reference instantiates (falcon-r1cs/src/gadgets/poly.rs:244,
# falcon-r1cs/examples/pok_sig.rs:3).
P_BLS12_381_FR = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


class SynthesisError(Exception):
    pass


# ---------------------------------------------------------------------------------------
# ark-relations::r1cs::{Variable, LinearCombination, ConstraintSystem}
# ---------------------------------------------------------------------------------------
# A Variable is a tuple (kind, index) with kind in: 'Z' zero, 'O' one, 'I' instance,
# 'W' witness, 'L' symbolic linear combination.
ZERO = ("Z", 0)
ONE = ("O", 0)


class ConstraintSystem:
    """ark_relations::r1cs::ConstraintSystem<F> in prove mode (construct_matrices = true)."""

    def __init__(self, p: int = P_BLS12_381_FR):
        self.p = p
        self.instance_assignment = [1]  # ConstraintSystem::new(): instance_assignment = vec![F::one()]
        self.witness_assignment = []
        self.lcs = []  # lc_map: index -> list[(coeff, Variable)]
        self.a = []
        self.b = []
        self.c = []
        self._lc_vals = []

    # --- counters ------------------------------------------------------------------
    def num_instance_variables(self):
        return len(self.instance_assignment)

    def num_witness_variables(self):
        return len(self.witness_assignment)

    def num_constraints(self):
        return len(self.a)

    def is_in_setup_mode(self):
        return False

    # --- allocation ----------------------------------------------------------------
    def new_input_variable(self, value: int):
        self.instance_assignment.append(value % self.p)
        return ("I", len(self.instance_assignment) - 1)

    def new_witness_variable(self, value: int):
        self.witness_assignment.append(value % self.p)
        return ("W", len(self.witness_assignment) - 1)

    def new_lc(self, lc):
        self.lcs.append(lc)
        return ("L", len(self.lcs) - 1)

    def enforce_constraint(self, a, b, c):
        # ark-relations stores each side as a fresh symbolic LC; only the count and the
        # satisfaction relation matter for the oracle.
        self.a.append(a)
        self.b.append(b)
        self.c.append(c)

    # --- evaluation ----------------------------------------------------------------
    def _eval_all_lcs(self):
        # A symbolic LC only ever refers to LCs created before it, so one forward pass in
        # creation order evaluates them all (ark-relations caches the same way, recursively).
        p = self.p
        inst, wit = self.instance_assignment, self.witness_assignment
        vals = []
        for lc in self.lcs:
            acc = 0
            for coeff, (kind, idx) in lc:
                if kind == "W":
                    acc += coeff * wit[idx]
                elif kind == "L":
                    acc += coeff * vals[idx]
                elif kind == "O":
                    acc += coeff
                elif kind == "I":
                    acc += coeff * inst[idx]
            vals.append(acc % p)
        self._lc_vals = vals

    def assigned_value(self, var):
        kind, idx = var
        if kind == "Z":
            return 0
        if kind == "O":
            return 1
        if kind == "I":
            return self.instance_assignment[idx]
        if kind == "W":
            return self.witness_assignment[idx]
        return self._lc_vals[idx]

    def eval_lc(self, lc):
        acc = 0
        for coeff, var in lc:
            acc += coeff * self.assigned_value(var)
        return acc % self.p

    def which_is_unsatisfied(self):
        """Index of the first unsatisfied constraint, or None (ConstraintSystem::which_is_unsatisfied)."""
        self._eval_all_lcs()
        p = self.p
        for i in range(len(self.a)):
            if (self.eval_lc(self.a[i]) * self.eval_lc(self.b[i]) - self.eval_lc(self.c[i])) % p:
                return i
        return None

    def is_satisfied(self):
        return self.which_is_unsatisfied() is None

    def to_matrices(self):
        """ConstraintSystem::to_matrices() after inline_all_lcs(): three lists of rows, each row a sorted list of
        (column, coeff) with column j < num_instance = instance j (0 = the constant one), else witness j - num_instance;
        duplicates summed, zero coefficients dropped."""
        p = self.p
        ni = len(self.instance_assignment)
        inl = []

        def expand(lc):
            acc = {}
            for coeff, (kind, idx) in lc:
                if kind == "Z":
                    continue
                if kind == "L":
                    for col, c in inl[idx]:
                        acc[col] = (acc.get(col, 0) + coeff * c) % p
                else:
                    col = 0 if kind == "O" else (idx if kind == "I" else ni + idx)
                    acc[col] = (acc.get(col, 0) + coeff) % p
            return sorted((col, c) for col, c in acc.items() if c)

        for lc in self.lcs:
            inl.append(expand(lc))
        return ([expand(r) for r in self.a], [expand(r) for r in self.b], [expand(r) for r in self.c])


# ---------------------------------------------------------------------------------------
# ark-r1cs-std::fields::fp::{AllocatedFp, FpVar}
# ---------------------------------------------------------------------------------------
class AllocatedFp:
    __slots__ = ("cs", "value", "variable")

    def __init__(self, cs, value, variable):
        self.cs = cs
        self.value = value % cs.p
        self.variable = variable

    # AllocatedFp::new_variable(mode = Witness | Input)
    @staticmethod
    def new_witness(cs, value):
        value %= cs.p
        return AllocatedFp(cs, value, cs.new_witness_variable(value))

    @staticmethod
    def new_input(cs, value):
        value %= cs.p
        return AllocatedFp(cs, value, cs.new_input_variable(value))

    # AllocatedFp::new_variable(mode = Constant): an LC (c, One); no variable allocated
    @staticmethod
    def new_constant(cs, c):
        return AllocatedFp(cs, c, cs.new_lc([(c % cs.p, ONE)]))

    def add(self, o):
        return AllocatedFp(self.cs, self.value + o.value,
                           self.cs.new_lc([(1, self.variable), (1, o.variable)]))

    def sub(self, o):
        return AllocatedFp(self.cs, self.value - o.value,
                           self.cs.new_lc([(1, self.variable), (self.cs.p - 1, o.variable)]))

    def mul(self, o):
        # AllocatedFp::mul: product witness, then a * b = product
        prod = AllocatedFp.new_witness(self.cs, self.value * o.value)
        self.cs.enforce_constraint([(1, self.variable)], [(1, o.variable)], [(1, prod.variable)])
        return prod

    def add_constant(self, c):
        c %= self.cs.p
        if c == 0:
            return self
        return AllocatedFp(self.cs, self.value + c, self.cs.new_lc([(1, self.variable), (c, ONE)]))

    def sub_constant(self, c):
        return self.add_constant(-c)

    def mul_constant(self, c):
        c %= self.cs.p
        return AllocatedFp(self.cs, self.value * c, self.cs.new_lc([(c, self.variable)]))

    def double(self):
        return AllocatedFp(self.cs, 2 * self.value,
                           self.cs.new_lc([(1, self.variable), (1, self.variable)]))

    def negate(self):
        return AllocatedFp(self.cs, -self.value, self.cs.new_lc([(self.cs.p - 1, self.variable)]))

    def is_neq(self, o):
        # AllocatedFp::is_neq: Boolean witness is_not_equal (with booleanity) + multiplier witness (no variable of
        # its own type: a bare cs.new_witness_variable), then
        #   (self - other) * multiplier = is_not_equal ;  (self - other) * not(is_not_equal) = 0
        cs = self.cs
        ne = self.value != o.value
        is_not_equal = Boolean.new_witness(cs, ne)
        mult = pow((self.value - o.value) % cs.p, cs.p - 2, cs.p) if ne else 1
        mvar = cs.new_witness_variable(mult)
        diff = [(1, self.variable), (cs.p - 1, o.variable)]
        cs.enforce_constraint(diff, [(1, mvar)], is_not_equal.lc())
        cs.enforce_constraint(diff, is_not_equal.not_().lc(), [])
        return is_not_equal

    def is_eq(self, o):
        return self.is_neq(o).not_()

    def conditional_enforce_equal(self, o, should_enforce):
        # (self - other) * should_enforce = 0
        self.cs.enforce_constraint(
            [(1, self.variable), (self.cs.p - 1, o.variable)], should_enforce.lc(), [])


class FpVar:
    """FpVar<F>: Constant(F) | Var(AllocatedFp<F>)."""
    __slots__ = ("const", "var")

    def __init__(self, const=None, var=None):
        self.const = const
        self.var = var

    @staticmethod
    def constant(cs, c):
        # FpVar::new_constant -> FpVar::Constant; allocates nothing
        return FpVar(const=c % cs.p)

    @staticmethod
    def new_witness(cs, value):
        return FpVar(var=AllocatedFp.new_witness(cs, value))

    @staticmethod
    def new_input(cs, value):
        return FpVar(var=AllocatedFp.new_input(cs, value))

    def is_constant(self):
        return self.var is None

    def value(self):
        return self.const if self.var is None else self.var.value

    # impl_ops!(FpVar, Add/Sub/Mul)
    def __add__(self, o):
        if self.var is None and o.var is None:
            return FpVar(const=self.const + o.const)
        if self.var is None:
            return FpVar(var=o.var.add_constant(self.const))
        if o.var is None:
            return FpVar(var=self.var.add_constant(o.const))
        return FpVar(var=self.var.add(o.var))

    def __sub__(self, o):
        if self.var is None and o.var is None:
            return FpVar(const=self.const - o.const)
        if o.var is None:
            return FpVar(var=self.var.sub_constant(o.const))
        if self.var is None:
            return FpVar(var=o.var.sub_constant(self.const).negate())
        return FpVar(var=self.var.sub(o.var))

    def __mul__(self, o):
        if self.var is None and o.var is None:
            return FpVar(const=self.const * o.const)
        if self.var is None:
            return FpVar(var=o.var.mul_constant(self.const))
        if o.var is None:
            return FpVar(var=self.var.mul_constant(o.const))
        return FpVar(var=self.var.mul(o.var))

    def double(self):
        if self.var is None:
            return FpVar(const=2 * self.const)
        return FpVar(var=self.var.double())

    @staticmethod
    def from_boolean(b):
        # impl From<Boolean<F>> for FpVar<F>
        if b.kind == "C":
            return FpVar(const=1 if b.const else 0)
        cs = b.ab.cs
        return FpVar(var=AllocatedFp(cs, 1 if b.value() else 0, cs.new_lc(b.lc())))

    def is_eq(self, o):
        # EqGadget::is_eq for FpVar: constants fold; a Constant operand is wrapped (new_constant) and becomes `self`
        if self.var is None and o.var is None:
            return Boolean.constant(self.const == o.const)
        if self.var is None:
            return AllocatedFp.new_constant(o.var.cs, self.const).is_eq(o.var)
        if o.var is None:
            return AllocatedFp.new_constant(self.var.cs, o.const).is_eq(self.var)
        return self.var.is_eq(o.var)

    def is_zero(self):
        # FieldVar::is_zero = self.is_eq(&Self::zero()); FpVar::is_eq with a Constant operand wraps the constant in
        # an AllocatedFp (new_constant) and calls c.is_eq(v): the constant is `self` of AllocatedFp::is_neq
        if self.var is None:
            return Boolean.constant(self.const == 0)
        c = AllocatedFp.new_constant(self.var.cs, 0)
        return c.is_eq(self.var)

    def enforce_equal(self, o):
        # EqGadget::enforce_equal -> conditional_enforce_equal(other, &Boolean::TRUE)
        t = Boolean.constant(True)
        if self.var is None and o.var is None:
            if self.const != o.const:
                raise SynthesisError("UnconstrainedVariable")
            return
        if self.var is None:
            AllocatedFp.new_constant(o.var.cs, self.const).conditional_enforce_equal(o.var, t)
        elif o.var is None:
            AllocatedFp.new_constant(self.var.cs, o.const).conditional_enforce_equal(self.var, t)
        else:
            self.var.conditional_enforce_equal(o.var, t)

    @staticmethod
    def conditionally_select(cond, true_value, false_value):
        # CondSelectGadget for FpVar / AllocatedFp
        if cond.kind == "C":
            return true_value if cond.const else false_value
        cs = cond.ab.cs
        if true_value.var is None and false_value.var is None:
            is_ = FpVar.from_boolean(cond).var
            not_ = FpVar.from_boolean(cond.not_()).var
            return FpVar(var=is_.mul_constant(true_value.const).add(not_.mul_constant(false_value.const)))
        tv = true_value.var if true_value.var is not None else AllocatedFp.new_constant(cs, true_value.const)
        fv = false_value.var if false_value.var is not None else AllocatedFp.new_constant(cs, false_value.const)
        result = AllocatedFp.new_witness(cs, tv.value if cond.value() else fv.value)
        # cond * (true - false) = result - false
        cs.enforce_constraint(
            cond.lc(),
            [(1, tv.variable), (cs.p - 1, fv.variable)],
            [(1, result.variable), (cs.p - 1, fv.variable)])
        return FpVar(var=result)


# ---------------------------------------------------------------------------------------
# ark-r1cs-std::bits::boolean::{AllocatedBool, Boolean}
# ---------------------------------------------------------------------------------------
class AllocatedBool:
    __slots__ = ("cs", "value", "variable")

    def __init__(self, cs, value, variable):
        self.cs = cs
        self.value = bool(value)
        self.variable = variable

    @staticmethod
    def new_witness(cs, value):
        # AllocatedBool::new_variable: witness, then booleanity (1 - a) * a = 0
        v = cs.new_witness_variable(1 if value else 0)
        cs.enforce_constraint([(1, ONE), (cs.p - 1, v)], [(1, v)], [])
        return AllocatedBool(cs, value, v)

    @staticmethod
    def _new_witness_without_booleanity_check(cs, value):
        return AllocatedBool(cs, value, cs.new_witness_variable(1 if value else 0))

    def or_(self, b):
        # (1 - a) * (1 - b) = (1 - result)
        cs = self.cs
        r = AllocatedBool._new_witness_without_booleanity_check(cs, self.value | b.value)
        cs.enforce_constraint([(1, ONE), (cs.p - 1, self.variable)],
                              [(1, ONE), (cs.p - 1, b.variable)],
                              [(1, ONE), (cs.p - 1, r.variable)])
        return r

    def and_(self, b):
        # a * b = result
        cs = self.cs
        r = AllocatedBool._new_witness_without_booleanity_check(cs, self.value & b.value)
        cs.enforce_constraint([(1, self.variable)], [(1, b.variable)], [(1, r.variable)])
        return r

    def and_not(self, b):
        # a * (1 - b) = result
        cs = self.cs
        r = AllocatedBool._new_witness_without_booleanity_check(cs, self.value & (not b.value))
        cs.enforce_constraint([(1, self.variable)], [(1, ONE), (cs.p - 1, b.variable)], [(1, r.variable)])
        return r

    def nor(self, b):
        # (1 - a) * (1 - b) = result
        cs = self.cs
        r = AllocatedBool._new_witness_without_booleanity_check(cs, (not self.value) & (not b.value))
        cs.enforce_constraint([(1, ONE), (cs.p - 1, self.variable)],
                              [(1, ONE), (cs.p - 1, b.variable)],
                              [(1, r.variable)])
        return r


class Boolean:
    """Boolean<F>: Is(AllocatedBool) | Not(AllocatedBool) | Constant(bool)."""
    __slots__ = ("kind", "ab", "const")

    def __init__(self, kind, ab=None, const=None):
        self.kind = kind  # 'I' | 'N' | 'C'
        self.ab = ab
        self.const = const

    @staticmethod
    def constant(b):
        return Boolean("C", const=bool(b))

    @staticmethod
    def new_witness(cs, value):
        return Boolean("I", ab=AllocatedBool.new_witness(cs, value))

    def value(self):
        if self.kind == "C":
            return self.const
        return self.ab.value if self.kind == "I" else (not self.ab.value)

    def lc(self):
        # Boolean::lc
        if self.kind == "C":
            return [(1, ONE)] if self.const else []
        if self.kind == "I":
            return [(1, self.ab.variable)]
        return [(1, ONE), (self.ab.cs.p - 1, self.ab.variable)]

    def not_(self):
        if self.kind == "C":
            return Boolean.constant(not self.const)
        return Boolean("N" if self.kind == "I" else "I", ab=self.ab)

    def xor_const(self, c: bool):
        # Boolean::xor with a Constant operand: x ^ false = x, x ^ true = !x  (no allocation)
        return self.not_() if c else self

    def is_eq_const(self, c: bool):
        # EqGadget::is_eq for Boolean = self.xor(other)?.not()
        return self.xor_const(c).not_()

    def and_(self, o):
        # Boolean::and
        if self.kind == "C":
            return o if self.const else Boolean.constant(False)
        if o.kind == "C":
            return self if o.const else Boolean.constant(False)
        if self.kind == "I" and o.kind == "N":
            return Boolean("I", ab=self.ab.and_not(o.ab))
        if self.kind == "N" and o.kind == "I":
            return Boolean("I", ab=o.ab.and_not(self.ab))
        if self.kind == "N" and o.kind == "N":
            return Boolean("I", ab=self.ab.nor(o.ab))
        return Boolean("I", ab=self.ab.and_(o.ab))

    def or_(self, o):
        # Boolean::or: any operand that is a Not is rewritten a OR b = NOT(NOT a AND NOT b)
        if self.kind == "C":
            return Boolean.constant(True) if self.const else o
        if o.kind == "C":
            return Boolean.constant(True) if o.const else self
        if self.kind == "I" and o.kind == "I":
            return Boolean("I", ab=self.ab.or_(o.ab))
        # match arms: (a @ Is, b @ Not) | (b @ Not, a @ Is) | (b @ Not, a @ Not)
        if self.kind == "I":
            a, b = self, o
        else:
            b, a = self, o
        return a.not_().and_(b.not_()).not_()

    @staticmethod
    def kary_and(bits):
        cur = None
        for nxt in bits:
            cur = nxt if cur is None else cur.and_(nxt)
        return cur

    @staticmethod
    def kary_or(bits):
        cur = None
        for nxt in bits:
            cur = nxt if cur is None else cur.or_(nxt)
        return cur

    def enforce_equal_const(self, c: bool):
        # EqGadget::conditional_enforce_equal(self, Constant(c), &Boolean::TRUE): difference * 1 = 0
        if self.kind == "C":
            if self.const != c:
                raise SynthesisError("AssignmentMissing")
            return
        cs = self.ab.cs
        v = self.ab.variable
        one_minus = [(1, ONE), (cs.p - 1, v)]
        if self.kind == "I":
            diff = one_minus if c else [(1, v)]
        else:
            diff = [(1, v)] if c else one_minus
        cs.enforce_constraint(diff, [(1, ONE)], [])
