"""Closed-form constraint rows for the device family SCO_FAM_STATE_PROGRAM (include/sco_hip.h).

The reference differentiates and evaluates arbitrary Python callables (`Expr(f)`, /root/reference/sco_py/expr.py:22-41);
a GPU cannot call Python.  Whatever can be written down with + - * / sin cos sqrt exp over the state x of a timestep and
a per-problem parameter vector p can be handed to the device as a small postfix program instead:

    from sco_py_amd.rowexpr import X, P, sin, sqrt, compile_rows
    rows = [P(2) - sqrt((X(0) - P(0)) ** 2 + (X(1) - P(1)) ** 2),       # keep out of a disc (centre, radius in p)
            X(1) - (0.3 * sin(2.0 * X(0)) + 0.8)]                        # stay below a wavy wall
    prog = compile_rows(rows)                     # .words, .row_ptr, .consts  ->  sco_sqp_load_program
    f = prog.numpy_fn(p)                          # the same rows as a NumPy callable x -> g(x): what Expr(f) gets

Both forms run the SAME operation sequence in double precision, so the host callable (reference, mirror API, oracle)
and the device program agree to the last bits of sin / cos / sqrt / exp.

r03: the state of a row may be the concatenation of ``span`` = 2 consecutive timesteps (X(i), i < 2 dof: the reference binds
any Expr to any Variable, expr.py:413-437 -- swept volumes, dynamics); the last ``n_eq`` rows of a block may be equalities
g(x, p) = 0 (EqExpr -> abs penalty, prob.py:280-315); one more expression may be a non-quadratic OBJECTIVE term of a timestep
(Prob.add_obj_expr on a plain Expr, prob.py:88-104); and ``Program.jacobian`` differentiates the rows in forward mode with
the same rules, in the same order of operations, as the device interpreter (csrc/sco_sqp.hip: prog_dual) -- the ``grad`` a
caller hands to ``Expr(f, grad)`` (expr.py:86-100).
"""
import numpy as np

OP_END, OP_X, OP_P, OP_C, OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_NEG, OP_SIN, OP_COS, OP_SQRT, OP_EXP, OP_SQUARE = range(14)
STACK = 16


class Node(object):
    """Expression tree node; build with X(i), P(k), numbers and the operators / functions below."""

    def __init__(self, op, arg=0, kids=(), const=None):
        self.op, self.arg, self.kids, self.const = op, arg, tuple(kids), const

    @staticmethod
    def lift(v):
        return v if isinstance(v, Node) else Node(OP_C, const=float(v))

    def __add__(self, o): return Node(OP_ADD, kids=(self, Node.lift(o)))
    def __radd__(self, o): return Node(OP_ADD, kids=(Node.lift(o), self))
    def __sub__(self, o): return Node(OP_SUB, kids=(self, Node.lift(o)))
    def __rsub__(self, o): return Node(OP_SUB, kids=(Node.lift(o), self))
    def __mul__(self, o): return Node(OP_MUL, kids=(self, Node.lift(o)))
    def __rmul__(self, o): return Node(OP_MUL, kids=(Node.lift(o), self))
    def __truediv__(self, o): return Node(OP_DIV, kids=(self, Node.lift(o)))
    def __rtruediv__(self, o): return Node(OP_DIV, kids=(Node.lift(o), self))
    def __neg__(self): return Node(OP_NEG, kids=(self,))

    def __pow__(self, k):
        if k != 2:
            raise ValueError("only ** 2 is available (write products for other powers)")
        return Node(OP_SQUARE, kids=(self,))

    def emit(self, words, consts):
        for k in self.kids:
            k.emit(words, consts)
        if self.op == OP_C:
            if self.const not in consts:
                consts.append(self.const)
            words.append((OP_C, consts.index(self.const)))
        else:
            words.append((self.op, self.arg))


def X(i): return Node(OP_X, int(i))
def P(k): return Node(OP_P, int(k))
def sin(a): return Node(OP_SIN, kids=(Node.lift(a),))
def cos(a): return Node(OP_COS, kids=(Node.lift(a),))
def sqrt(a): return Node(OP_SQRT, kids=(Node.lift(a),))
def exp(a): return Node(OP_EXP, kids=(Node.lift(a),))


class Program(object):
    """Rows compiled to the wire format of sco_sqp_load_program."""

    def __init__(self, words, row_ptr, consts, n_eq=0, span=1, objective=False):
        self.words = np.ascontiguousarray(words, dtype=np.int32).reshape(-1, 2)
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
        self.consts = np.ascontiguousarray(consts, dtype=np.float64)
        self.objective = bool(objective)           # the LAST program is an objective term, not a constraint row
        self.n_rows = len(self.row_ptr) - 1 - (1 if self.objective else 0)      # constraint rows of a block
        self.n_eq = int(n_eq)                      # how many of them (the last ones) are equalities
        self.span = int(span)                      # timesteps per constraint block
        ops = self.words
        self.n_params = int(ops[ops[:, 0] == OP_P, 1].max()) + 1 if np.any(ops[:, 0] == OP_P) else 0
        self.n_state = int(ops[ops[:, 0] == OP_X, 1].max()) + 1 if np.any(ops[:, 0] == OP_X) else 0

    def evaluate(self, x, p=(), rows=None):
        """g (n_rows,) at state x with parameters p: the interpreter the device runs, in NumPy scalars."""
        x = np.asarray(x, dtype=np.float64).ravel(); p = np.asarray(p, dtype=np.float64).ravel()
        rows = range(self.n_rows) if rows is None else list(rows)
        res = np.zeros(len(rows))
        out = {}
        for r in rows:
            st = []
            for op, arg in self.words[self.row_ptr[r]:self.row_ptr[r + 1] - 1]:
                if op == OP_X: st.append(np.float64(x[arg]))
                elif op == OP_P: st.append(np.float64(p[arg]))
                elif op == OP_C: st.append(np.float64(self.consts[arg]))
                elif op == OP_NEG: st[-1] = -st[-1]
                elif op == OP_SIN: st[-1] = np.sin(st[-1])
                elif op == OP_COS: st[-1] = np.cos(st[-1])
                elif op == OP_SQRT: st[-1] = np.sqrt(st[-1])
                elif op == OP_EXP: st[-1] = np.exp(st[-1])
                elif op == OP_SQUARE: st[-1] = st[-1] * st[-1]
                else:
                    b = st.pop(); a = st[-1]
                    st[-1] = a + b if op == OP_ADD else a - b if op == OP_SUB else a * b if op == OP_MUL else a / b
            out[r] = st[0]
        for k, r in enumerate(rows):
            res[k] = out[r]
        return res

    def jacobian(self, x, p=(), rows=None):
        """d g / d x (len(rows), len(x)) by forward-mode differentiation: one pass per coordinate j with (value, derivative)
        pairs on the stack -- the rules and their order of operations are those of the device (prog_dual)."""
        x = np.asarray(x, dtype=np.float64).ravel(); p = np.asarray(p, dtype=np.float64).ravel()
        rows = range(self.n_rows) if rows is None else list(rows)
        J = np.zeros((len(rows), x.shape[0]))
        one, zero, two = np.float64(1.0), np.float64(0.0), np.float64(2.0)
        for k, r in enumerate(rows):
            for j in range(x.shape[0]):
                sv, sd = [], []
                for op, arg in self.words[self.row_ptr[r]:self.row_ptr[r + 1] - 1]:
                    if op == OP_X: sv.append(np.float64(x[arg])); sd.append(one if arg == j else zero)
                    elif op == OP_P: sv.append(np.float64(p[arg])); sd.append(zero)
                    elif op == OP_C: sv.append(np.float64(self.consts[arg])); sd.append(zero)
                    elif op == OP_NEG: sv[-1] = -sv[-1]; sd[-1] = -sd[-1]
                    elif op == OP_SIN: sd[-1] = np.cos(sv[-1]) * sd[-1]; sv[-1] = np.sin(sv[-1])
                    elif op == OP_COS: sd[-1] = -(np.sin(sv[-1]) * sd[-1]); sv[-1] = np.cos(sv[-1])
                    elif op == OP_SQRT: rt = np.sqrt(sv[-1]); sd[-1] = sd[-1] / (two * rt); sv[-1] = rt
                    elif op == OP_EXP: ev = np.exp(sv[-1]); sd[-1] = ev * sd[-1]; sv[-1] = ev
                    elif op == OP_SQUARE: sd[-1] = (two * sv[-1]) * sd[-1]; sv[-1] = sv[-1] * sv[-1]
                    else:
                        bv, bd = sv.pop(), sd.pop()
                        av, ad = sv[-1], sd[-1]
                        if op == OP_ADD: sv[-1] = av + bv; sd[-1] = ad + bd
                        elif op == OP_SUB: sv[-1] = av - bv; sd[-1] = ad - bd
                        elif op == OP_MUL: sd[-1] = ad * bv + av * bd; sv[-1] = av * bv
                        else: qv = av / bv; sd[-1] = (ad - qv * bd) / bv; sv[-1] = qv
                J[k, j] = sd[0]
        return J

    def numpy_fn(self, p=(), rows=None):
        """x -> g(x) for fixed parameters: the callable to wrap in the reference's / the mirror's Expr(f)."""
        p = np.array(p, dtype=np.float64)
        return lambda x: self.evaluate(x, p, rows)

    def numpy_jac(self, p=(), rows=None):
        """x -> dg/dx for fixed parameters: the ``grad`` of Expr(f, grad)."""
        p = np.array(p, dtype=np.float64)
        return lambda x: self.jacobian(x, p, rows)

    # the rows of a block by kind, and the objective term (index n_rows)
    @property
    def ineq_rows(self): return list(range(self.n_rows - self.n_eq))

    @property
    def eq_rows(self): return list(range(self.n_rows - self.n_eq, self.n_rows))

    def objective_fn(self, p=()):
        """x -> f(x) (a float): the objective term of a timestep (needs ``objective=``)."""
        assert self.objective
        p = np.array(p, dtype=np.float64)
        return lambda x: float(self.evaluate(x, p, rows=[self.n_rows])[0])


def compile_rows(rows, eq_rows=(), objective=None, span=1):
    """List of Node expressions -> Program.  ``rows``: inequality rows g_r(x, p) <= 0; ``eq_rows``: equality rows
    g_r(x, p) = 0 (they follow the inequalities in a block); ``objective``: a non-quadratic objective term f(x, p) of ONE
    timestep (span 1 only); ``span``: timesteps per constraint block (X(i) addresses i < span * dof)."""
    if span not in (1, 2, 3, 4) or (objective is not None and span != 1):
        raise ValueError("span is 1 .. 4; an objective term needs span 1")
    rows, eq_rows = list(rows), list(eq_rows)          # (generators are welcome: they are walked once, here)
    words, row_ptr, consts = [], [0], []
    for r in rows + eq_rows + ([objective] if objective is not None else []):
        Node.lift(r).emit(words, consts)
        words.append((OP_END, 0))
        row_ptr.append(len(words))
    prog = Program(words, row_ptr, consts, n_eq=len(eq_rows), span=span, objective=objective is not None)
    for r in range(len(prog.row_ptr) - 1):            # the same checks the C side makes, with Python errors
        sp = 0
        for op, arg in prog.words[prog.row_ptr[r]:prog.row_ptr[r + 1] - 1]:
            if op in (OP_X, OP_P, OP_C): sp += 1
            elif OP_ADD <= op <= OP_DIV: sp -= 1
            if sp > STACK:
                raise ValueError("row %d needs a deeper stack than %d" % (r, STACK))
    return prog
