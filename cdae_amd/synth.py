"""Synthetic implicit-feedback datasets shaped like the BASELINE.json configs.

The reference trains on a "user item" text file that is not in the repository
(/root/reference/apps/yelp/yelp.cpp:23, 60-66) and splits it per user with the first floor(0.2*n)
shuffled instances going to test (/root/reference/src/base/data-inl.hpp:231-272).  There is no
network here, so bench.py and the tests generate data of the same shape:

  * user activity n_u ~ log-normal, clipped to [min_items, I/4], mean set by `nnz`
  * item popularity Zipf(s); 20 % of every user's draws come from a global ranking, 80 % from the
    ranking of the user's latent group (a per-group permutation of the items), so that a model which
    learns user structure beats the popularity baseline by a wide margin (Recall@10 ~0.23 vs ~0.09 at
    4000 x 1500) and Recall@10 is a meaningful parity signal
  * ids dense 0..U-1 / 0..I-1, rows sorted ascending and unique
  * per-user split: floor(test_ratio * n) random items to test, the rest to train

Everything is numpy-vectorised and deterministic in `seed` (default echoes yelp.cpp:29).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

SHAPES = {
    # name: (users, items, total nnz)  — SURVEY.md §8 table
    "ml10m": (70_000, 10_600, 10_000_000),
    "netflix": (480_000, 17_700, 100_000_000),
    "yelp": (10_000, 7_000, 312_500),
    # BASELINE configs[4] item space (1 M items, ~100 interactions per user) with as many users as a single-GPU bench needs
    "cfg5_items": (20_000, 1_000_000, 2_000_000),
    # reduced configs[4] for the parity fixture: > 65 536 items (32-bit sort keys, 256-row GEMM tiles) at K = 512
    "cfg5_small": (256, 131_072, 25_600),
    # accuracy envelope of the K = 512 full-output block schedule (tools/accuracy_envelope.py --full-output): the smallest item space
    # that takes configs[4]'s launches (>= 32768 items: GEMM 3 fused with the row step) with enough users for Recall@10 to mean something
    "cfg5_env": (16_384, 32_768, 1_638_400),
    "tiny": (300, 120, 9_000),
    "small": (4_000, 1_500, 240_000),
}


@dataclass
class Interactions:
    num_users: int
    num_items: int
    train_ptr: np.ndarray  # int64 [U+1]
    train_col: np.ndarray  # uint32 [nnz_train], ascending inside a row
    test_ptr: np.ndarray
    test_col: np.ndarray

    @property
    def nnz_train(self) -> int:
        return int(self.train_ptr[-1])

    def user_range(self, u0: int, u1: int) -> "Interactions":
        """Rows [u0, u1) re-based to 0 — the shard a data-parallel rank trains on."""
        def cut(ptr, col):
            p = ptr[u0:u1 + 1] - ptr[u0]
            return p.astype(np.int64), col[ptr[u0]:ptr[u1]].copy()
        tp, tc = cut(self.train_ptr, self.train_col)
        sp, sc = cut(self.test_ptr, self.test_col)
        return Interactions(u1 - u0, self.num_items, tp, tc, sp, sc)


def _csr_from_pairs(users: np.ndarray, items: np.ndarray, num_users: int):
    key = users.astype(np.int64) * (1 << 32) + items.astype(np.int64)
    key = np.unique(key)  # sorted by (user, item), duplicates dropped
    u = (key >> 32).astype(np.int64)
    i = (key & 0xFFFFFFFF).astype(np.uint32)
    ptr = np.zeros(num_users + 1, dtype=np.int64)
    np.add.at(ptr, u + 1, 1)
    np.cumsum(ptr, out=ptr)
    return ptr, i


def generate(num_users: int, num_items: int, nnz: int, seed: int = 20141119, zipf_s: float = 1.0,
             groups: int = 16, min_items: int = 20, test_ratio: float = 0.2,
             group_mix: float = 0.8) -> Interactions:
    rng = np.random.default_rng(seed)
    max_items = max(min_items + 1, num_items // 4)
    mean_target = nnz / num_users
    sigma = 0.8
    mu = np.log(max(mean_target, 1.0)) - 0.5 * sigma * sigma
    n_u = np.clip(np.round(rng.lognormal(mu, sigma, num_users)), min_items, max_items).astype(np.int64)
    # rescale once so the clipped mean lands near the target
    n_u = np.clip(np.round(n_u * (mean_target / n_u.mean())), min_items, max_items).astype(np.int64)

    ranks = np.arange(1, num_items + 1, dtype=np.float64)
    pmf = ranks ** (-zipf_s)
    cdf = np.cumsum(pmf / pmf.sum())
    # group g sees the global ranking through its own permutation
    perms = np.stack([rng.permutation(num_items) for _ in range(groups)]).astype(np.uint32)
    user_group = rng.integers(0, groups, num_users)

    over = 1.7
    draws = np.ceil(n_u * over).astype(np.int64) + 8
    owner = np.repeat(np.arange(num_users, dtype=np.int64), draws)
    r = np.searchsorted(cdf, rng.random(owner.size), side="right").clip(0, num_items - 1)
    from_group = rng.random(owner.size) < group_mix
    items = np.where(from_group, perms[user_group[owner], r], r).astype(np.uint32)
    # order of arrival inside a user decides which unique items survive the n_u cut
    arrival = rng.random(owner.size)
    order = np.lexsort((arrival, owner))
    owner, items = owner[order], items[order]
    key = owner * (1 << 32) + items
    _, first = np.unique(key, return_index=True)
    keep = np.zeros(owner.size, dtype=bool)
    keep[first] = True
    owner, items = owner[keep], items[keep]       # still in (owner, arrival) order
    start = np.searchsorted(owner, np.arange(num_users))
    pos_in_user = np.arange(owner.size) - start[owner]
    sel = pos_in_user < n_u[owner]
    owner, items, pos_in_user = owner[sel], items[sel], pos_in_user[sel]
    cnt = np.bincount(owner, minlength=num_users)
    n_test = np.floor(test_ratio * cnt).astype(np.int64)   # data-inl.hpp:252
    is_test = pos_in_user < n_test[owner]                    # arrival order is already random
    tr_ptr, tr_col = _csr_from_pairs(owner[~is_test], items[~is_test], num_users)
    te_ptr, te_col = _csr_from_pairs(owner[is_test], items[is_test], num_users)
    assert (np.diff(tr_ptr) >= 1).all(), "every user needs a train item (cdae.hpp:139)"
    return Interactions(num_users, num_items, tr_ptr, tr_col, te_ptr, te_col)


def generate_shape(name: str, seed: int = 20141119, **kw) -> Interactions:
    u, i, nnz = SHAPES[name]
    return generate(u, i, nnz, seed=seed, **kw)


def generate_uniform(num_users: int, num_items: int, per_user: int = 20, seed: int = 20141119, test_ratio: float = 0.2) -> Interactions:
    """Scale runs only (bench.py --users above 2 M: BASELINE configs[4]'s 10 M users on one box): every user has exactly `per_user`
    interactions, one uniform draw from each of `per_user` equal strata of the item ids (unique and ascending by construction, no
    sort, no hashing: 10 M users take seconds and a few GB); the first floor(test_ratio * per_user) of a fixed pseudo-random
    column order go to test.  No popularity skew, no group structure: NOT an accuracy workload."""
    rng = np.random.default_rng(seed)
    width = num_items // per_user
    assert width >= 1
    items = (np.arange(per_user, dtype=np.uint32) * np.uint32(width))[None, :] + rng.integers(0, width, (num_users, per_user), dtype=np.uint32)
    n_test = int(np.floor(test_ratio * per_user))
    cols = np.random.default_rng(seed + 1).permutation(per_user)
    test_cols, train_cols = np.sort(cols[:n_test]), np.sort(cols[n_test:])
    tr_col = np.ascontiguousarray(items[:, train_cols]).reshape(-1)
    te_col = np.ascontiguousarray(items[:, test_cols]).reshape(-1)
    tr_ptr = np.arange(num_users + 1, dtype=np.int64) * train_cols.size
    te_ptr = np.arange(num_users + 1, dtype=np.int64) * test_cols.size
    return Interactions(num_users, num_items, tr_ptr, tr_col, te_ptr, te_col)

