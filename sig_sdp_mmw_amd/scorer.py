"""Scoring of a colouring: per-user SINR and block error rate.

Restatement of `env.evaluate_sinr` / `env.evaluate_bler` (sim_src/env/env.py:198-236) -- the step every
reference driver runs after the hot path (e.g. sim_script/pd_mmw_template.py:31-32).  The reference
densifies a K x K gain matrix and loops over users; here the interference a user suffers is read off
per-slot member-by-member blocks of the K x A receive-power matrix (O(K^2/Z) numbers in total, never K x K at once).
"""
import math

import numpy as np
import scipy.spatial.distance

from .graphs import _NOISE_FLOOR_DBM, _bler_polyanskiy, min_sinr_dec

__all__ = ["receive_power", "evaluate_sinr", "evaluate_bler"]


def receive_power(sta_locs, ap_locs, fre_Hz=4e9, txp_offset=2.0):
    """Unthresholded receive powers rx[k, a] (env._compute_state_real, env.py:157-166)."""
    dis = scipy.spatial.distance.cdist(sta_locs, ap_locs)
    L0 = 20.0 * math.log10(fre_Hz / 1e6) + 16 - 28
    loss = L0 + 28 * np.log10(dis + 1)
    smax = np.max(-loss, axis=1)
    t = 10.0 * math.log10(min_sinr_dec()) - (smax - _NOISE_FLOOR_DBM)
    txp = np.reshape(t + 10.0 * math.log10(txp_offset), (sta_locs.shape[0], -1))
    return 10 ** ((txp - loss - _NOISE_FLOOR_DBM) / 10.0)


def evaluate_sinr(rx, z, Z):
    """SINR per user under colouring `z` (env.py:198-225).  rx: K x A receive powers."""
    K = rx.shape[0]
    z = np.asarray(z).astype(np.int64)
    asso = np.argmax(rx, axis=1)
    signal = rx[np.arange(K), asso]
    sinr = np.zeros(K) + 1e-3
    for zz in range(Z):
        mem = np.nonzero(z == zz)[0]
        if mem.size == 0:
            continue
        # T[i, j] = power of member j at member i's AP, own link removed; summed in the reference's order
        # (a row sum of the dense slot sub-matrix, env.py:210) so that the near-ties the power control creates
        # between users of one AP resolve the same way in the collision rule below
        T = np.asfortranarray(rx[np.ix_(mem, asso[mem])].T)  # the reference's sub-matrix is column-major: members accumulate one by one
        np.fill_diagonal(T, 0.0)
        sinr[mem] = signal[mem] / (T.sum(axis=1) + 1)
    # users of one AP colliding in a slot: only the strongest survives (env.py:214-224)
    key = asso.astype(np.int64) * Z + z
    order = np.lexsort((-sinr, key))
    first = np.ones(K, dtype=bool)
    first[1:] = key[order][1:] != key[order][:-1]
    inside = (z >= 0) & (z < Z)
    losers = order[~first]
    sinr[losers[inside[losers]]] = 1e-3
    return sinr


def evaluate_bler(rx, z, Z, packet_bit=800, bandwidth=5e6, slot_time=1.25e-4):
    """Block error rate per user (env.py:227-233, finite-blocklength model env.py:107-111)."""
    sinr = evaluate_sinr(rx, z, Z)
    return np.array([_bler_polyanskiy(s, packet_bit, bandwidth, slot_time) for s in sinr])
