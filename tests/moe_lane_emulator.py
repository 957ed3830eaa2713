"""A LANE-LEVEL emulator of the reference's grouped_topk_sigmoid kernel (csrc/kernel/moe/grouped_topk_sigmoid_kernel.cu:64-180):
THRS_PER_TOKEN lanes (one per expert group, :194), each holding its ELTS_PER_THR scores, and the literal `__shfl_xor_sync`
butterflies with their tie rules — "higher indices win" when the group with the smallest top-2 sum is looked for
(:110-121), "lower indices win" in the arg-max of the top-k rounds (:149-161).  fp32 arithmetic throughout (numpy).

Why it exists (round-5 review, item 7a): the reference offers nothing to pin this function against — its Python model only
implements the "greedy" gate — so oracle/moe.py and hx_grouped_topk_sigmoid used to be held to each other only.  This is a
second, independent restatement at the level where the tie-breaks live; tests hold BOTH to it on tie-heavy inputs for the
four (experts, groups) pairs the reference instantiates (:233-268).  It is still a restatement: the row stays
"parity unpinned"."""
import numpy as np

FLT_MAX = np.float32(np.finfo(np.float32).max)


def _butterfly(vals, cols, n_lanes, better):
    """The xor butterfly of :110-121 / :149-161 over one token's lanes: every lane ends with the winner."""
    vals, cols = list(vals), list(cols)
    mask = n_lanes // 2
    while mask > 0:
        nv, nc = list(vals), list(cols)
        for lane in range(n_lanes):          # all lanes exchange at once: read the OLD values of the partner
            ov, oc = vals[lane ^ mask], cols[lane ^ mask]
            if better(ov, oc, vals[lane], cols[lane]):
                nv[lane], nc[lane] = ov, oc
        vals, cols = nv, nc
        mask //= 2
    return vals, cols


def grouped_topk_sigmoid_lanes(scores, bias, n_groups, topk_group, topk):
    """scores: fp32 [n_tokens, n_experts] = sigmoid(logits) as the caller's implementation computes it (the emulation
    starts behind :81-84's expf); bias fp32 [n_experts].  Returns (weights fp32 [n_tokens, topk], indices int32)."""
    scores = np.asarray(scores, dtype=np.float32)
    bias = np.asarray(bias, dtype=np.float32)
    n_tokens, n_experts = scores.shape
    T = n_groups                                  # THRS_PER_TOKEN = EXPERTS_GROUPS (:194)
    E = n_experts // T                            # ELTS_PER_THR
    assert T & (T - 1) == 0 and T <= 32 and n_experts == T * E
    weights = np.zeros((n_tokens, topk), dtype=np.float32)
    indices = np.zeros((n_tokens, topk), dtype=np.int32)
    for t in range(n_tokens):
        sc = [scores[t, l * E:(l + 1) * E].copy() for l in range(T)]                       # scores_chunk per lane
        tmp = [(scores[t, l * E:(l + 1) * E] + bias[l * E:(l + 1) * E]).astype(np.float32) for l in range(T)]   # :83
        start_col = [l * E for l in range(T)]
        # 2. mask out the (T - topk_group) groups with the smallest top-2 sums (:88-131)
        for _ in range(T - topk_group):
            sums = []
            for l in range(T):
                mx = sm = -FLT_MAX
                for v in tmp[l]:
                    if v > mx:
                        sm, mx = mx, v
                    elif v > sm:
                        sm = v
                with np.errstate(over="ignore"):
                    sums.append(np.float32(mx + sm))                     # fp32: FLT_MAX + FLT_MAX = +inf for a masked group
            _, cols = _butterfly(sums, start_col, T, lambda ov, oc, v, c: ov < v or (ov == v and oc > c))
            assert len(set(cols)) == 1                                   # every lane agrees on the group to clear
            tmp[cols[0] // E][:] = FLT_MAX
        # 3. top-k over what is left (:134-176)
        for k in range(topk):
            vals, cols = [], []
            for l in range(T):
                mx, col = tmp[l][0], start_col[l]
                if mx != FLT_MAX:
                    for i in range(1, E):
                        if tmp[l][i] > mx:
                            mx, col = tmp[l][i], start_col[l] + i
                else:
                    mx = -FLT_MAX
                vals.append(mx)
                cols.append(col)
            _, cols = _butterfly(vals, cols, T, lambda ov, oc, v, c: ov > v or (ov == v and oc < c))
            assert len(set(cols)) == 1
            col = cols[0]
            tmp[col // E][col % E] = -FLT_MAX
            weights[t, k] = sc[col // E][col % E]
            indices[t, k] = col
    return weights, indices


INSTANTIATED = [(128, 4), (128, 8), (256, 8), (256, 16)]          # grouped_topk_sigmoid_kernel.cu:233-268


def tie_heavy_inputs(n_experts, n_tokens, seed):
    """Logits and biases from small sets of values: equal logits give equal sigmoids in every implementation, dyadic
    biases add exactly — ties between experts and between groups' top-2 sums are everywhere."""
    rng = np.random.RandomState(seed)
    logits = rng.choice(np.array([-1.0, 0.0, 0.5, 2.0], dtype=np.float32), size=(n_tokens, n_experts))
    bias = rng.choice(np.array([0.0, 0.125, -0.25, 0.5], dtype=np.float32), size=(n_experts,))
    logits[0] = 0.0                      # one token with every score equal
    if n_tokens > 1:
        logits[1] = logits[1, 0]         # ... and one whose groups differ by the bias alone
    return logits, bias
