"""Problem instances for the MMW hot path: `state = (S_gain, Q_asso, h_max)`.

Two families, both seeded and reproducible from NumPy alone (nothing here touches the GPU; the
solver consumes the scipy CSR triples exactly as the reference's solver does):

* ``journal_graph``  – restatement of the reference's problem generator
  (sim_src/env/env.py:16-59 station drop, :93-97 path loss, :136-143 power control,
  :145-155 thresholded receive power, :168-196 S/Q/h_max assembly) in vectorised sparse form.
  The reference builds ``Q`` with a Python loop over APs on a LIL matrix (env.py:181-183) and
  densifies K x A twice; here ``Q = M M^T - I`` and ``S = R M^T`` with ``M`` the K x A
  association indicator, which is the same matrix.
* ``er_contention_graph`` – the synthetic Erdos-Renyi family SURVEY.md §8(d) defines for the
  benchmark configs (the reference has no synthetic generator).

`S_gain[k, j]` is the interference user k causes at user j's access point (row k = what k
emits), the diagonal is the own-link gain; `Q_asso` is the symmetric 0/1 same-AP relation with a
zero diagonal; `h_max[k]` the tolerable interference of user k.
"""
import math

import numpy as np
import scipy.sparse
import scipy.spatial.distance
import scipy.stats

__all__ = ["journal_graph", "journal_graph_device", "journal_geometry", "er_contention_graph", "min_sinr_dec", "instance_stats"]

_NOISE_FLOOR_DBM = -94.0  # env.py:9


def _bler_polyanskiy(snr_dec, L, B, T):
    # finite-blocklength error model, env.py:107-111
    nu = -L * math.log(2.0) + B * T * math.log(1 + snr_dec)
    do = math.sqrt(B * T * (1.0 - 1.0 / ((1.0 + snr_dec) ** 2)))
    return scipy.stats.norm.sf(nu / do)


def min_sinr_dec(packet_bit=800, bandwidth=5e6, slot_time=1.25e-4, max_err=1e-5):
    """SINR (linear) at which the block error rate meets ``max_err`` (env.py:113-134, bisection in dB)."""

    def err(x_db):
        return _bler_polyanskiy(10.0 ** (x_db / 10.0), packet_bit, bandwidth, slot_time) / max_err - 1.0

    a, b, tol = -5.0, 30.0, 0.1
    if err(a) * err(b) >= 0:
        raise ValueError("bisection bracket does not contain the target error rate")
    while (err(a) - err(b)) > tol:
        mid = (a + b) / 2
        e_mid = err(mid)
        if e_mid == 0:
            return 10.0 ** (mid / 10.0)
        if err(a) * e_mid < 0:
            b = mid
        else:
            a = mid
    return 10.0 ** (((a + b) / 2) / 10.0)


def journal_graph(cell_size=20, sta_density_per_1m2=5e-3, seed=1, cell_edge=20.0, fre_Hz=4e9, txp_offset=2.0,
                  min_s_n_ratio=0.1, return_geometry=False):
    """``env(cell_size, sta_density_per_1m2, seed).generate_S_Q_hmax()`` restated (env.py:168-196).

    APs sit on a ``cell_size x cell_size`` grid of pitch ``cell_edge``; ``K = int(cell_size**2 *
    rho * cell_edge**2)`` users are dropped uniformly with ``default_rng(seed).uniform``; each user's
    transmit power is set so its strongest AP receives ``txp_offset * min_sinr`` over the noise floor;
    receive powers below ``min_s_n_ratio`` are dropped.
    """
    grid_edge = cell_edge * cell_size
    n_ap = int(cell_size ** 2)
    n_sta = int(cell_size ** 2 * (sta_density_per_1m2 * cell_edge ** 2))
    off = cell_edge / 2.0
    ax = np.linspace(0 + off, grid_edge - off, cell_size)
    xx, yy = np.meshgrid(ax, ax)
    ap_locs = np.array((xx.ravel(), yy.ravel())).T
    sta_locs = np.random.default_rng(seed).uniform(low=0.0, high=grid_edge, size=(n_sta, 2))

    dis = scipy.spatial.distance.cdist(sta_locs, ap_locs)
    L0 = 20.0 * math.log10(fre_Hz / 1e6) + 16 - 28
    loss = L0 + 28 * np.log10(dis + 1)
    msinr = min_sinr_dec()
    smax = np.max(-loss, axis=1)
    t = 10.0 * math.log10(msinr) - (smax - _NOISE_FLOOR_DBM)
    txp = np.reshape(t + 10.0 * math.log10(txp_offset), (n_sta, -1))
    rx_db = txp - loss - _NOISE_FLOOR_DBM
    rx = 10 ** (rx_db / 10.0)
    rx[rx < min_s_n_ratio] = 0.0

    asso = np.argmax(rx, axis=1)
    K = n_sta
    M = scipy.sparse.csr_matrix((np.ones(K), (np.arange(K), asso)), shape=(K, n_ap))
    Q = (M @ M.T).tocsr()
    Q.setdiag(0.0)
    Q.eliminate_zeros()
    Q.sort_indices()
    Q.data[:] = 1.0
    R = scipy.sparse.csr_matrix(rx)
    S = (R @ M.T).tocsr()
    S.eliminate_zeros()
    S.sort_indices()
    h_max = S.diagonal() / msinr - 1.0
    if return_geometry:
        return (S, Q, h_max), {"sta_locs": sta_locs, "ap_locs": ap_locs, "asso": asso}
    return S, Q, h_max


def journal_geometry(cell_size=20, sta_density_per_1m2=5e-3, seed=1, cell_edge=20.0):
    """AP grid and station drop of `env.__init__` (env.py:16-59): (sta_locs (K, 2), ap_locs (A, 2))."""
    grid_edge = cell_edge * cell_size
    n_sta = int(cell_size ** 2 * (sta_density_per_1m2 * cell_edge ** 2))
    off = cell_edge / 2.0
    ax = np.linspace(0 + off, grid_edge - off, cell_size)
    xx, yy = np.meshgrid(ax, ax)
    ap_locs = np.array((xx.ravel(), yy.ravel())).T
    sta_locs = np.random.default_rng(seed).uniform(low=0.0, high=grid_edge, size=(n_sta, 2))
    return sta_locs, ap_locs


def journal_graph_device(cell_size=20, sta_density_per_1m2=5e-3, seed=1, cell_edge=20.0, fre_Hz=4e9, txp_offset=2.0, min_s_n_ratio=0.1,
                         device=0):
    """The same instance as `journal_graph`, generated on the GPU (`mmw_env_create`, csrc/env_device.h): the host only draws the
    2 K station coordinates.  Returns (state, env) where `env` also scores colourings (`env.evaluate(z, Z)`)."""
    from . import _lib
    sta_locs, ap_locs = journal_geometry(cell_size, sta_density_per_1m2, seed, cell_edge)
    env = _lib.DeviceEnv(sta_locs, ap_locs, fre_Hz=fre_Hz, txp_offset=txp_offset, min_s_n_ratio=min_s_n_ratio, min_sinr=min_sinr_dec(),
                         noise_floor_dbm=_NOISE_FLOOR_DBM, device=device)
    return env.state(), env


def er_contention_graph(K, p, seed, clique=3, lo=0.1, hi=3.7, own_gain=None):
    """Synthetic contention instance, SURVEY.md §8(d): directed off-diagonal Bernoulli(p) mask, gains
    log-uniform on [lo, hi], diagonal ``own_gain`` (default ``hi``), ``Q`` = disjoint ``clique``-cliques
    over a random permutation of the users (~3 users per AP as in the journal generator), ``h_max = 1``.
    """
    rng = np.random.default_rng(seed)
    own_gain = hi if own_gain is None else own_gain
    if p >= 1.0:
        rows = np.repeat(np.arange(K), K)
        cols = np.tile(np.arange(K), K)
        keep = rows != cols
        rows, cols = rows[keep], cols[keep]
    else:
        # draw the number of off-diagonal entries per row, then distinct columns per row
        n_off = rng.binomial(K - 1, p, size=K)
        rows = np.repeat(np.arange(K), n_off)
        cols = np.empty(rows.size, dtype=np.int64)
        pos = 0
        for k in range(K):
            c = rng.choice(K - 1, size=n_off[k], replace=False)
            c = np.where(c >= k, c + 1, c)  # skip the diagonal
            cols[pos:pos + n_off[k]] = c
            pos += n_off[k]
    vals = np.exp(rng.uniform(math.log(lo), math.log(hi), size=rows.size))
    S = scipy.sparse.csr_matrix((np.concatenate([vals, np.full(K, own_gain)]),
                                 (np.concatenate([rows, np.arange(K)]), np.concatenate([cols, np.arange(K)]))),
                                shape=(K, K))
    S.sort_indices()
    perm = rng.permutation(K)
    grp = np.empty(K, dtype=np.int64)
    grp[perm] = np.arange(K) // clique
    M = scipy.sparse.csr_matrix((np.ones(K), (np.arange(K), grp)), shape=(K, int(grp.max()) + 1))
    Q = (M @ M.T).tocsr()
    Q.setdiag(0.0)
    Q.eliminate_zeros()
    Q.sort_indices()
    Q.data[:] = 1.0
    h_max = np.ones(K)
    return S, Q, h_max


def instance_stats(state):
    """Sizes in the notation of SURVEY.md §8: K, directed gain nnz, E_Q, C."""
    S, Q, _ = state
    K = S.shape[0]
    return {"K": K, "S_nnz": int(S.nnz), "E_Q": int(Q.nnz // 2), "C": int(Q.nnz // 2 + 2 * K),
            "density": float(S.nnz) / float(K) / float(K)}
