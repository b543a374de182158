"""CPU ORACLE for the MMW hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (sig_sdp_mmw_amd) never does: it runs on hand-written HIP kernels through the C-ABI in
include/mmw_hip.h and fails loudly when that library is missing.

What this is: a NumPy/SciPy restatement of the reference's solver path
(`/root/reference/sim_src/alg/mmw.py`, `sim_src/alg/sdp_solver.py:18-107`) *as executed*, written
on a fixed sparsity pattern with flat value arrays (O(nnz) per iteration) instead of the
reference's scipy.sparse matrix algebra.  Each function cites the reference lines it follows.

Pinned: tests/test_oracle_golden.py checks every function here against tests/golden/*.npz, which
tests/golden/make_golden.py captured from the reference itself (imported in the build container,
numpy 2.2.6 / scipy 1.15.3).  The reference's own tests hold no vectors for this path.

Third-party arithmetic the reference delegates to SciPy, and that this oracle delegates to the same
SciPy calls (same image on the GPU box): `scipy.sparse.linalg.expm_multiply` (mmw.py:228,
Al-Mohy & Higham 2011), `scipy.sparse.linalg.svds` (mmw.py:215, ARPACK), `eigsh` (mmw.py:115).

As-executed quirks kept:
  * S_T' = S_gain^T lives in CSC, so `csr_scal_rows_inplace` (scipy_util.py:20-24) scales COLUMNS;
  * the DUAL phase multiplies S_T' by X_offdi as matrices and row-sums (mmw.py:133-134), i.e.
    eH_raw = S_T' (X_offdi 1), not the Hadamard contraction.
"""
import math

import numpy as np
import scipy.sparse
import scipy.sparse.linalg

__all__ = ["Pattern", "process_state", "softmax", "expm_half", "sketch_rows", "MMWOracle", "factor_xavg",
           "rounding_one_attempt", "rounding_one_attempt_as_executed", "rounding", "projector"]


# --------------------------------------------------------------------------------------------
# A3 / A4: state processing and the fixed pattern
# --------------------------------------------------------------------------------------------
def process_state(Z, S_gain, Q_asso, h_max):
    """mmw._process_state, mmw.py:26-41.  Returns (ST as CSR of the K x K matrix S_T', S_sum, norm_H).

    S_T'[k, j] = S_gain[j, k] with the diagonal and every position in nz(Q) removed.
    """
    K = S_gain.shape[0]
    ST = scipy.sparse.csr_matrix(S_gain).T.tocsr().astype(np.float64)  # values S_gain[j,k] at (k,j)
    ST.sort_indices()
    coo = ST.tocoo()
    Qb = scipy.sparse.csr_matrix(Q_asso)
    qmask = scipy.sparse.csr_matrix((np.ones(Qb.nnz, dtype=bool) & (Qb.data != 0), Qb.indices, Qb.indptr), shape=Qb.shape)
    in_q = np.asarray(qmask[coo.row, coo.col]).ravel() if coo.nnz else np.zeros(0, dtype=bool)
    keep = (~in_q) & (coo.row != coo.col) & (coo.data != 0)
    ST = scipy.sparse.csr_matrix((coo.data[keep], (coo.row[keep], coo.col[keep])), shape=(K, K))
    ST.sort_indices()
    # row sums accumulate in ascending column order (csc @ ones), mmw.py:34
    S_sum = np.zeros(K)
    sq = np.zeros(K)
    rows = np.repeat(np.arange(K), np.diff(ST.indptr))
    order = np.argsort(ST.indices, kind="stable")  # ascending column, rows arbitrary but each row's entries stay col-sorted
    np.add.at(S_sum, rows[order], ST.data[order])
    np.add.at(sq, rows[order], ST.data[order] ** 2)
    norm_H = np.sqrt(sq) * (Z - 1) / (2 * Z) + np.abs(1 / K * h_max - 1 / K / Z * S_sum)  # mmw.py:39
    return ST, S_sum, norm_H


class Pattern:
    """Fixed index structure of one (state, Z) solve: mmw.py:49-68 plus the L/X pattern of mmw.py:144-194."""

    def __init__(self, Z, state):
        S_gain, Q_asso, h_max = state
        self.K = K = S_gain.shape[0]
        self.Z = Z
        self.h_max = np.asarray(h_max, dtype=np.float64)
        self.ST, self.S_sum, self.norm_H = process_state(Z, S_gain, Q_asso, self.h_max)
        self.cH = 1.0 / K * self.h_max - 1 / (K * Z) * self.S_sum  # mmw.py:167
        # upper-triangular edge lists in CSR row-major order, mmw.py:52-57
        sym = (self.ST + self.ST.T).tocsr()
        up = scipy.sparse.triu(sym).tocsr()
        up.eliminate_zeros()
        up.sort_indices()
        self.gain_x, self.gain_y = (a.astype(np.int64) for a in up.nonzero())
        qu = scipy.sparse.triu(scipy.sparse.csr_matrix(Q_asso), k=1).tocsr()
        qu.sort_indices()
        self.asso_x, self.asso_y = (a.astype(np.int64) for a in qu.nonzero())
        self.E_asso = int(scipy.sparse.csr_matrix(Q_asso).getnnz() / 2)  # mmw.py:59
        self.E_gain = self.gain_x.size
        self.C = self.E_asso + 2 * K
        # L pattern = diag + sym(ST pattern) + Q pattern
        rows = np.concatenate([np.arange(K), self.gain_x, self.gain_y, self.asso_x, self.asso_y])
        cols = np.concatenate([np.arange(K), self.gain_y, self.gain_x, self.asso_y, self.asso_x])
        key = np.unique(rows * K + cols)
        assert key.size == rows.size, "gain and association edges must be disjoint"
        self.nnzL = key.size
        self.row = (key // K).astype(np.int64)
        self.col = (key % K).astype(np.int64)
        self.indptr = np.concatenate([[0], np.cumsum(np.bincount(self.row, minlength=K))]).astype(np.int64)
        self._key = key
        self.diag_pos = self.pos(np.arange(K), np.arange(K))
        self.gain_ab, self.gain_ba = self.pos(self.gain_x, self.gain_y), self.pos(self.gain_y, self.gain_x)
        self.asso_ab, self.asso_ba = self.pos(self.asso_x, self.asso_y), self.pos(self.asso_y, self.asso_x)
        # per undirected gain edge (a<b): S_T'[a,b] and S_T'[b,a]
        self.w_ab = np.asarray(self.ST[self.gain_x, self.gain_y]).ravel() if self.E_gain else np.zeros(0)
        self.w_ba = np.asarray(self.ST[self.gain_y, self.gain_x]).ravel() if self.E_gain else np.zeros(0)

    def pos(self, r, c):
        p = np.searchsorted(self._key, np.asarray(r, dtype=np.int64) * self.K + np.asarray(c, dtype=np.int64))
        return p.astype(np.int64)

    def csr(self, vals):
        return scipy.sparse.csr_matrix((np.asarray(vals, dtype=np.float64), self.col.astype(np.int32), self.indptr.astype(np.int32)),
                                       shape=(self.K, self.K))


# --------------------------------------------------------------------------------------------
# per-phase arithmetic
# --------------------------------------------------------------------------------------------
def softmax(x):
    """scipy.special.softmax as called at mmw.py:139: exp(x - max) / sum."""
    e = np.exp(x - np.max(x))
    return e / np.sum(e)


def sketch_rows(g):
    """mmw.expm_half_randsk, mmw.py:226-227: (K,D) normals -> rows of unit 2-norm."""
    D = g.shape[1]
    r = g / math.sqrt(float(D))
    return r / np.linalg.norm(r, axis=1)[:, None]


def expm_half(L_half_csr, randv):
    """mmw.py:228: the action of exp(L/2) on the sketch (SciPy's Al-Mohy-Higham truncated Taylor)."""
    return scipy.sparse.linalg.expm_multiply(scipy.sparse.csr_matrix(L_half_csr).copy(), randv)


def violations(p, xval, dual_as_executed=False):
    """e(X) of mmw.py:124-137 for X given by its values on the pattern (diag included)."""
    K, Z = p.K, p.Z
    xd = xval[p.diag_pos]
    eD = (xd - 1.0) / (1.0 - 1.0 / K)
    eF = (xval[p.asso_ab] + 1.0 / (Z - 1)) / (1.0 / (K * (Z - 1)) + 1.0 / 2.0)
    if dual_as_executed:
        xo = np.array(xval, dtype=np.float64)
        xo[p.diag_pos] = 0.0
        X_offdi = p.csr(xo)
        X_offdi.eliminate_zeros()
        AHX = p.ST.tocsc() * X_offdi  # sparse x sparse product, mmw.py:133
        raw = np.asarray(AHX.sum(axis=1)).ravel()
    else:
        r = np.bincount(p.row, weights=xval, minlength=K) - xd  # X_offdi 1
        raw = p.ST @ r
    eH = (raw * (Z - 1) / Z - (p.h_max - (1 / Z * p.S_sum))) / p.norm_H
    return np.hstack((eD, eF, eH))


def loss_values(p, Y):
    """(LD + LF + LH) of mmw.py:146-165 as values on the pattern."""
    K, Z = p.K, p.Z
    YD, YF, YH = Y[0:K], Y[K:K + p.E_asso], Y[K + p.E_asso:2 * K + p.E_asso]
    out = np.zeros(p.nnzL)
    cF = 1.0 / 2.0 + 1.0 / (K * (Z - 1))
    w = YH * (1.0 / p.norm_H)  # column scaling of the CSC matrix, scipy_util.py:20-24 via mmw.py:162-163
    diag = (YD - np.sum(YD) / K) / (1.0 - 1.0 / K)
    diag = diag + (np.sum(YF) / (K * (Z - 1))) / cF
    diag = diag - np.sum(p.cH * YH / p.norm_H)
    out[p.diag_pos] = diag
    f = (YF / 2.0) / cF
    out[p.asso_ab] = f
    out[p.asso_ba] = f
    g = (p.w_ab * w[p.gain_y] + p.w_ba * w[p.gain_x]) * (Z - 1) / (2 * Z)
    out[p.gain_ab] = g
    out[p.gain_ba] = g
    return out


def x_on_pattern(p, X_half):
    """mmw.py:182-194: X = X_half X_half^T / (tr/K) sampled on the pattern."""
    d = np.sum(X_half * X_half, axis=1)
    tr = np.sum(d) / p.K
    xval = np.zeros(p.nnzL)
    xval[p.diag_pos] = d / tr
    g = np.einsum("ij,ij->i", X_half[p.gain_x], X_half[p.gain_y]) / tr
    xval[p.gain_ab] = g
    xval[p.gain_ba] = g
    a = np.einsum("ij,ij->i", X_half[p.asso_x], X_half[p.asso_y]) / tr
    xval[p.asso_ab] = a
    xval[p.asso_ba] = a
    return xval


def factor_xavg(Xavg_csr, rank, v0=None):
    """mmw.py:213-216: top-`rank` singular triplets of the averaged X, X_half = U sqrt(S)."""
    u, s, vT = scipy.sparse.linalg.svds(Xavg_csr, k=rank, v0=v0)
    return np.matmul(u, np.diag(np.sqrt(s)))


def projector(X_half):
    """Sign/rotation-invariant view of a factor: X_half X_half^T."""
    return X_half @ X_half.T


class MMWOracle:
    """The MMW loop of mmw._run (mmw.py:44-222) on flat arrays.

    `sketch(i, K, D)` must return the (K, D) row-normalised sketch of iteration i (parity mode:
    the recorded draws; baseline mode: `sketch_rows(np.random.randn(K, D))` like mmw.py:226).

    `keep_trace`: True keeps every iteration's quantities, a collection of iteration indices only those (long
    full-size runs).  Besides the per-iteration values the trace holds, per kept iteration i, the running sums
    mmw.py:77-78 hold when iteration i + 1 starts ("xsum" / "ysum": X_0 + ... + X_{i+1}, same for Y).
    """

    def __init__(self, nit=100, rank_radio=2, eta=0.1, log_gap=False, dual_as_executed=False, expm=expm_half):
        self.nit, self.rank_radio, self.eta, self.log_gap = nit, rank_radio, eta, log_gap
        self.dual_as_executed = dual_as_executed
        self.expm = expm
        self.trace = None

    def run(self, Z, state, sketch, keep_trace=False, factor=True, v0=None, pattern=None):
        p = Pattern(Z, state) if pattern is None else pattern
        K, C, eta = p.K, p.C, self.eta
        D = Z * self.rank_radio
        Y = np.ones(C) / C
        e_accu = np.zeros(C)
        lval = np.zeros(p.nnzL)
        xval = np.zeros(p.nnzL)
        xval[p.diag_pos] = 1.0
        xavg = np.zeros(p.nnzL)
        yavg = np.zeros(C)
        tr = {"e_this": [], "e_accu": [], "Y": [], "lval": [], "xval": [], "X_half": [], "gap": [], "xsum": [], "ysum": [], "iters": []}
        keep_all = keep_trace is True
        keep_set = set() if isinstance(keep_trace, bool) else set(int(i) for i in keep_trace)
        for i in range(self.nit):
            xavg += xval
            yavg += Y
            if self.log_gap:  # mmw.py:79-117
                n = i + 1
                e_max = np.max(violations(p, xavg / n, self.dual_as_executed))
                Lm = p.csr(loss_values(p, yavg / n))
                s, _ = scipy.sparse.linalg.eigsh(Lm, k=1, which="SA")
                tr["gap"].append([e_max, s[0] * K, e_max - s[0] * K])
            e_this = violations(p, xval, self.dual_as_executed)
            e_accu = e_accu + e_this * eta
            Y = softmax(e_accu)
            lval = lval - loss_values(p, Y) * eta
            X_half = self.expm(p.csr(lval / 2.0), sketch(i, K, D))
            xval = x_on_pattern(p, X_half)
            if keep_all or i in keep_set:
                tr["iters"].append(i)
                tr["xsum"].append(xavg + xval)
                tr["ysum"].append(yavg + Y)
                tr["e_this"].append(e_this)
                tr["e_accu"].append(e_accu.copy())
                tr["Y"].append(Y)
                tr["lval"].append(lval.copy())
                tr["xval"].append(xval)
                tr["X_half"].append(X_half)
        self.pattern = p
        self.trace = tr
        self.xavg = xavg / self.nit
        self.yavg = yavg / self.nit
        if not factor:
            return True, None
        rank = int(np.min([K - 1, (Z - 1) * self.rank_radio]))
        return True, factor_xavg(p.csr(self.xavg), rank, v0=v0)


# --------------------------------------------------------------------------------------------
# A13: rounding
# --------------------------------------------------------------------------------------------
def _out_lists(S_gain):
    S = scipy.sparse.csr_matrix(S_gain).copy()
    S.setdiag(0)
    S.eliminate_zeros()
    S.sort_indices()
    return S


def rounding_one_attempt(Z, gX, state, randv, randint=None):
    """sdp_solver.rounding_one_attempt (sdp_solver.py:27-107) with the draw `randv` (Z, D', already
    row-normalised as at :48-49) injected.  O(deg) per probe: the reference's dense row scans
    (:81,:89,:94-95) only ever touch neighbours, and its association test reduces to "no member of the
    slot shares an AP with k" (SURVEY.md §8a A13).  Returns (z_vec f64[K], Z, remainder, unassigned mask).
    """
    S_gain, Q_asso, h_max = state
    K = S_gain.shape[0]
    S = _out_lists(S_gain)
    Q = scipy.sparse.csr_matrix(Q_asso)
    order = np.argsort(-np.linalg.norm(gX, axis=1))  # :51
    inprod = np.matmul(randv, gX.transpose())  # :56
    pref = np.argsort(-inprod, axis=0)  # :57
    z_vec = np.zeros(K)
    slot = np.full(K, -1, dtype=np.int64)
    gain_sum = np.zeros((Z, K))
    for k in order:
        nb = S.indices[S.indptr[k]:S.indptr[k + 1]]
        gv = S.data[S.indptr[k]:S.indptr[k + 1]]
        qn = Q.indices[Q.indptr[k]:Q.indptr[k + 1]]
        nb_slot = slot[nb]
        qn_slot = slot[qn]
        for z in pref[:, k]:
            m = nb_slot == z
            if gain_sum[z, k] + 0.0 > h_max[k]:  # n = k, tmp_h[k] = 0 (:78-84)
                continue
            if np.any(gain_sum[z, nb[m]] + gv[m] > h_max[nb[m]]):
                continue
            if np.any(qn_slot == z):  # asso_sum[z][k] >= 1 (:86-92)
                continue
            gain_sum[z, nb] += gv  # :94, assignment order = accumulation order
            slot[k] = z
            z_vec[k] = z
            break
    un = slot < 0
    if np.any(un):
        fn = np.random.randint if randint is None else randint
        z_vec[un] = fn(Z, size=int(un.sum()))  # :104-105
    return z_vec, Z, int(un.sum()), un


def rounding_one_attempt_as_executed(Z, gX, state, randv, randint=None):
    """The same attempt following the reference's data flow literally (dense K-vectors per slot, the
    `np.split(Q.indices, S.indptr)` index lists of sdp_solver.py:40-41 included).  Small K only."""
    S_gain, Q_asso, h_max = state
    K = S_gain.shape[0]
    S = _out_lists(S_gain)
    Q = scipy.sparse.csr_matrix(Q_asso)
    S_idx = np.split(S.indices, S.indptr)[1:-1]
    Q_idx = np.split(Q.indices, S.indptr)[1:-1]
    Sd = np.asarray(S.todense())
    Qd = np.asarray(Q.todense())
    free = np.ones(K, dtype=bool)
    order = np.argsort(-np.linalg.norm(gX, axis=1))
    pref = np.argsort(-np.matmul(randv, gX.transpose()), axis=0)
    z_vec = np.zeros(K)
    gsum = [np.zeros(K) for _ in range(Z)]
    asum = [np.zeros(K) for _ in range(Z)]
    members = [[] for _ in range(Z)]
    for k in order:
        for z in pref[:, k]:
            nbr = np.append(np.intersect1d(np.array(members[z]), S_idx[k]), k).astype(int)
            if np.any((gsum[z][nbr] + Sd[k][nbr]) > h_max[nbr]):
                continue
            nbr = np.append(np.intersect1d(np.array(members[z]), Q_idx[k]), k).astype(int)
            if np.any((asum[z][nbr] + Qd[k][nbr]) >= 1):
                continue
            gsum[z] += Sd[k]
            asum[z] += Qd[k]
            members[z].append(k)
            free[k] = False
            z_vec[k] = z
            break
    if np.any(free):
        fn = np.random.randint if randint is None else randint
        z_vec[free] = fn(Z, size=int(free.sum()))
    return z_vec, Z, int(free.sum()), free


def rounding(Z, gX, state, draw_randv, randint=None, nattempt=10):
    """sdp_solver.rounding, sdp_solver.py:18-25: up to `nattempt` attempts, stop at remainder 0.
    `draw_randv(Z, D)` supplies each attempt's row-normalised projection vectors."""
    z_vec = rem = None
    for _ in range(nattempt):
        z_vec, Z, rem, _ = rounding_one_attempt(Z, gX, state, draw_randv(Z, gX.shape[1]), randint)
        if rem == 0:
            break
    return z_vec, Z, rem


def draw_randv_host(Z, D):
    """sdp_solver.py:48-49 on the global NumPy stream."""
    r = np.random.randn(Z, D)
    return r / np.linalg.norm(r, axis=1, keepdims=True)
