"""Detection-list comparison used by the parity tests and smoke() (TEST INFRASTRUCTURE).

Two fp32 implementations of the same network agree on scores to ~1e-6, so two detections whose
reference scores are closer than that may legitimately trade places in the sorted top-K list
(torch.topk itself leaves the order of ties unspecified, decode.py:106,112), and a detection
sitting on the K-th score may fall in or out.  ``match_rows`` therefore pairs every produced
row with the reference row it corresponds to -- same content within ``tol``, at most ``window``
ranks away -- and the callers assert, on the PAIRED rows, bit-identical indices / classes and
1e-4 boxes, that displaced pairs are near-ties, and that unpaired rows sit on the K-th score.
"""
import numpy as np


def match_rows(got, ref, cols, tol, window=6):
    """got, ref: (K, D) rows sorted by score.  Returns match (K,) int: index of the reference row
    paired with got[r], or -1.  Greedy in rank order, each reference row used once."""
    K = got.shape[0]
    used = np.zeros(ref.shape[0], bool)
    match = np.full(K, -1, np.int64)
    for r in range(K):
        lo, hi = max(0, r - window), min(ref.shape[0], r + window + 1)
        order = sorted(range(lo, hi), key=lambda j: abs(j - r))
        for j in order:
            if used[j]:
                continue
            if np.all(np.abs(got[r, cols] - ref[j, cols]) <= tol):
                match[r] = j
                used[j] = True
                break
    return match


def compare_topk(got, ref, score_col=4, box_cols=(0, 1, 2, 3), tie=2e-6, score_tol=1e-4,
                 box_tol=1e-4, got_ids=None, ref_ids=None, window=6):
    """Batch comparison.  got, ref: (B, K, D).  ``*_ids``: optional (B, K) integer identities
    (flat index, class ...) that must be IDENTICAL on paired rows.

    The rule (north_star: "bit-identical box indices / scores within 1e-4"): with eps = the image's
    largest score difference on paired rows (itself held to ``score_tol``), two detections can only
    trade ranks when the ORACLE's own scores at the two ranks lie within gap = max(tie, 2 eps) of each
    other -- so
      * at every SAFE rank (oracle score separated from both neighbours by more than gap) the row must
        sit in place, paired, with identical ids;
      * a displaced pair must be a near-tie: |s_ref[j] - s_ref[r]| <= gap * (|j - r| + 1) (a run of
        near-ties |j - r| + 1 ranks long);
      * an unpaired row may only be a boundary trade: its score within gap of the oracle's K-th.
    Raises AssertionError otherwise.  Returns the fractions paired / in_place / safe (share of ranks the
    strict rule covers) and the largest eps seen."""
    B, K, _ = got.shape
    cols = list(box_cols) + [score_col]
    scale = max(1.0, float(np.abs(ref[..., list(box_cols)]).max()))
    tol = np.array([box_tol * scale] * len(box_cols) + [score_tol])
    paired = in_place = n_safe = 0
    eps_max = 0.0
    for b in range(B):
        m = match_rows(got[b], ref[b], cols, tol, window)
        s_ref = ref[b, :, score_col].astype(np.float64)
        ok = m >= 0
        eps = float(np.abs(got[b, ok, score_col].astype(np.float64) - s_ref[m[ok]]).max()) if ok.any() else 0.0
        eps_max = max(eps_max, eps)
        gap = max(tie, 2.0 * eps)
        d_prev = np.abs(np.diff(s_ref, prepend=np.inf))
        d_next = np.abs(np.diff(s_ref, append=-np.inf))
        # the last rank also neighbours the (unseen) K+1-th oracle score: never counted as safe
        safe = (d_prev > gap) & (d_next > gap)
        safe[K - 1] = False
        n_safe += int(safe.sum())
        for r in range(K):
            j = m[r]
            if safe[r]:
                assert j == r, ("a rank the oracle separates by more than %.1e is not in place" % gap, b, r, int(j),
                                float(got[b, r, score_col]), float(s_ref[r]))
            if j < 0:
                # may only happen at the selection boundary: the row's score ties the K-th one
                assert abs(float(got[b, r, score_col]) - float(s_ref[K - 1])) <= gap, \
                    ("unpaired row off the boundary", b, r, float(got[b, r, score_col]), float(s_ref[K - 1]))
                continue
            paired += 1
            if j == r:
                in_place += 1
            else:
                assert abs(float(s_ref[j]) - float(s_ref[r])) <= gap * (abs(j - r) + 1), \
                    ("displaced pair is not a near-tie", b, r, j, float(s_ref[j]), float(s_ref[r]))
            if got_ids is not None:
                assert np.array_equal(got_ids[b, r], ref_ids[b, j]), ("identity differs", b, r, j)
    n = float(B * K)
    return {"paired": paired / n, "in_place": in_place / n, "safe": n_safe / n, "eps": eps_max}
