"""The fp64 arbiter of near-tied full-dims decisions (tests/test_fulldims_gpu.py; round 6) -- test infrastructure on top of oracle/.

Two fp32 evaluations of a K = 20 beam search that sum in different orders may rank near-tied candidates differently (profiles/r05_margin_search.txt:
~400 adjacent gaps per user, the smallest of the size of the re-association noise).  Neither is the judge of the other: both approximate the same
search in exact arithmetic, which `RefLlama(dtype=float64)` + `beamsd_ref.SCORE_DTYPE = float64` evaluate -- the SAME fp32 weight values widened per use."""
import torch

from oracle import beamsd_ref as R
from oracle.llama_ref import RefLlama


def arbiter_of(rt: RefLlama) -> RefLlama:
    """The fp64 view of an fp32 oracle model (cached on it; shares its weight tensors: no second copy of a 7B model)."""
    if not hasattr(rt, "_arbiter64"):
        rt._arbiter64 = RefLlama(rt.d, rt.w, max_slots=rt.kcache.shape[1], dtype=torch.float64)
    return rt._arbiter64


def fp64_truth(rt: RefLlama, prompt, K: int, fn):
    """The arbiter's list: the plain constrained beam search of the target (beamSD.py:544-595) in double precision -- by the lossless property
    (beam-SD == target_generate in exact arithmetic: every fixture of the real reference has it) the top-K any faithful evaluation approximates; a
    third of the tokens of an fp64 beam-SD run and no draft forwards.  -> (items [K][L], scores [K] as floats)"""
    R.SCORE_DTYPE = torch.float64
    try:
        truth = R.target_generate(arbiter_of(rt), prompt, 4, K, fn)
    finally:
        R.SCORE_DTYPE = torch.float32
    P = len(prompt)
    return truth["beam_sequence"][:, P:].tolist(), [float(x) for x in truth["beam_scores"]]


def fp64_gap(rt: RefLlama, prompt, items, t_items, t_sc) -> float:
    """Largest |fp64 score of the list's item at rank i - the fp64 search's score at rank i| (0 for a list that IS the fp64 list: no forward needed)."""
    if items == t_items:
        return 0.0
    sc64 = oracle_scores_of(arbiter_of(rt), prompt, items, dtype=torch.float64)
    return max(abs(a - b) for a, b in zip(sc64, t_sc))


def oracle_scores_of(ref_model, prompt, seqs, dtype=torch.float32):
    """oracle beam scores (fp32; fp64 with the arbiter model and dtype=torch.float64) of arbitrary generated sequences (sum of full-vocabulary
    log-probabilities, beamSD.py:58,69-70) from one packed forward: the prompt once, every sequence a branch under a tree mask."""
    P, L, n = len(prompt), len(seqs[0]), len(seqs)
    ids = [int(t) for t in prompt] + [int(t) for sq in seqs for t in sq[:-1]]
    T = len(ids)
    pos = list(range(P)) + [P + j for _ in seqs for j in range(L - 1)]
    vis = torch.zeros(T, T, dtype=torch.bool)
    vis[:P, :P] = torch.tril(torch.ones(P, P, dtype=torch.bool))
    for i in range(n):
        lo = P + i * (L - 1)
        vis[lo: lo + L - 1, :P] = True
        vis[lo: lo + L - 1, lo: lo + L - 1] = torch.tril(torch.ones(L - 1, L - 1, dtype=torch.bool))
    logp = torch.log_softmax(ref_model.forward(torch.tensor(ids), torch.tensor(pos), torch.arange(T), vis, n_logit_rows=T - P + 1).to(dtype), dim=-1)
    return [float(sum(logp[r, int(t)] for r, t in zip([0] + [1 + i * (L - 1) + j for j in range(L - 1)], sq))) for i, sq in enumerate(seqs)]
