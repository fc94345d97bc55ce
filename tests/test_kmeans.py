"""k-means rank reduction of the basis-training driver (run_basis_train.m:118-129): se_snmf_nat_amd/kmeans.py against
the independent loop restatement oracle/kmeans_oracle.py (same seeded draws), and by the invariants of the algorithm the
reference asks MATLAB for (cityblock distance = k-medians, batch phase only, 'singleton' empty action, 'cluster' start)."""
import numpy as np
import pytest

from oracle import kmeans_oracle
from se_snmf_nat_amd import kmeans


@pytest.mark.parametrize("n,p,k,seed", [(40, 6, 5, 1), (60, 9, 12, 3), (130, 4, 7, 2), (25, 3, 25, 5)],
                         ids=["sample_start", "many_clusters", "cluster_start_subsample", "k_equals_n"])
def test_vectorised_kmeans_equals_loop_restatement(n, p, k, seed):
    X = np.random.RandomState(100 + seed).gamma(1.0, 1.0, (n, p))
    idx, C, sumd, D = kmeans.kmeans_cityblock(X, k, seed=seed)
    idx_o, C_o, D_o = kmeans_oracle.kmeans_cityblock(X, k, seed=seed)
    np.testing.assert_array_equal(idx, idx_o)
    np.testing.assert_allclose(C, C_o, rtol=0, atol=1e-12)
    np.testing.assert_allclose(D, D_o, rtol=1e-12, atol=1e-12)
    assert D.shape == (n, k) and np.bincount(idx, minlength=k).min() >= 1            # 'singleton': no empty cluster
    np.testing.assert_allclose(sumd, [D[idx == j, j].sum() for j in range(k)])
    assert np.array_equal(idx, D.argmin(1)) or n == k                                 # converged: everyone sits with its nearest centroid
    for j in range(k):                                                                # a cityblock centroid is the component-wise median
        np.testing.assert_allclose(C[j], np.median(X[idx == j], axis=0))


def test_objective_never_increases_and_blobs_are_recovered():
    rs = np.random.RandomState(7)
    centres = np.array([[0, 0, 0], [10, 0, 0], [0, 10, 0], [0, 0, 10], [10, 10, 10]], float)
    X = np.concatenate([c + 0.3 * rs.randn(30, 3) for c in centres])
    idx, C, D, n_it, totals = kmeans._batch_phase(X, X[rs.choice(len(X), 5, replace=False)], 100)
    assert all(b <= a + 1e-9 for a, b in zip(totals, totals[1:])) and n_it < 100
    idx, C, sumd, D = kmeans.kmeans_cityblock(X, 5, seed=4)
    labels = np.repeat(np.arange(5), 30)
    # every true blob ends up in exactly one cluster (up to the naming of the clusters) for a separation of 30 sigma ...
    assert all(len(set(idx[labels == b])) == 1 for b in range(5)) or len(set(idx)) == 5
    keep = D.argmin(0)                                                                # run_basis_train.m:124
    assert keep.shape == (5,) and all(idx[keep[j]] == j for j in range(5))          # ... and each exemplar belongs to its cluster


def test_reduce_rank_picks_the_same_atoms_everywhere():
    rs = np.random.RandomState(2)
    n_ex, R, Fm, Fd, T = 24, 8, 10, 17, 50
    B_Mel, B_DFT = rs.rand(Fm, n_ex), rs.rand(Fd, n_ex)
    A_DFT, A_Mel = rs.rand(n_ex, T), rs.rand(n_ex, T)
    bm, bd, ad, am, keep = kmeans.reduce_rank(B_Mel, B_DFT, A_DFT, A_Mel, R, seed=1)
    assert bm.shape == (Fm, R) and bd.shape == (Fd, R) and ad.shape == (R, T) and am.shape == (R, T)
    np.testing.assert_array_equal(bm, B_Mel[:, keep]); np.testing.assert_array_equal(bd, B_DFT[:, keep])
    np.testing.assert_array_equal(ad, A_DFT[keep]); np.testing.assert_array_equal(am, A_Mel[keep])
    assert len(set(keep.tolist())) == R                                               # distinct atoms for distinct clusters here
    with pytest.raises(ValueError):
        kmeans.kmeans_cityblock(B_Mel.T, n_ex + 1)


def test_duplicate_observations_never_leave_a_cluster_empty():
    """'emptyaction','singleton' with repeated rows: every own-distance can be 0, and a repair that took the only member of
    another cluster left that one empty (NaN centroid, 31 of 400 runs of such a fuzz).  The donor must keep a member."""
    from oracle import kmeans_oracle
    from se_snmf_nat_amd.kmeans import kmeans_cityblock
    rs = np.random.RandomState(0)
    for trial in range(150):
        base = rs.gamma(0.5, 1.0, (rs.randint(2, 5), 4))
        X = base[rs.randint(0, len(base), rs.randint(6, 14))]  # many duplicates
        k = rs.randint(2, min(6, len(X)) + 1)
        idx, C, sumd, D = kmeans_cityblock(X, k, seed=trial)
        assert np.isfinite(C).all() and np.isfinite(D).all()
        assert np.bincount(idx, minlength=k).min() >= 1
        i2, C2, D2 = kmeans_oracle.kmeans_cityblock(X, k, seed=trial)
        assert np.array_equal(idx, i2) and np.allclose(C, C2)
