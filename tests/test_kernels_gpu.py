"""GPU parity of the individual HIP kernels, called through the C-ABI (ctypes), against the
oracle / plain fp32 torch on the same seeded inputs.  Run with -m gpu on the MI355X box."""
import ctypes as C


import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from atspeed_amd import _lib, synth
from atspeed_amd.generation_trie import PositionSetConstraint, Trie
from oracle import beamsd_ref as R


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    l = _lib.load()
    assert l.atspeed_device_count() >= 1
    return l


def _st():
    return _lib.stream_ptr()


def _rand(shape, seed, std=1.0):
    """Seeded N(0, std^2) fp32 values (CPU tensor).  Large operands come from the library's fill kernel, which test_fill_hash_normal_bit_exact pins
    bit for bit to synth.hash_normal (numpy): the 90 M-element weights of the Llama-7B shapes cost ~1.5 s each in numpy -- a third of the suite."""
    n = int(np.prod(shape))
    if n >= (1 << 20) and torch.cuda.is_available():
        t = torch.empty(n, dtype=torch.float32, device="cuda")
        _lib.check(_lib.load().atspeed_fill_hash_normal(t.data_ptr(), n, seed, float(synth.normal_scale(std)), 0.0, _lib.ATSPEED_F32, 0, _st()))
        return t.cpu().reshape(shape)
    return torch.from_numpy(synth.hash_normal(n, seed, std).reshape(shape))


# ------------------------------------------------------------------ fill
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_fill_hash_normal_bit_exact(lib, dtype):
    n, seed, std = 100003, 12345, 0.02
    t = torch.empty(n, dtype=dtype, device="cuda")
    code = _lib.dtype_code(dtype)
    _lib.check(lib.atspeed_fill_hash_normal(t.data_ptr(), n, seed, float(synth.normal_scale(std)), 0.0, code, 0, _st()))
    ref = synth.hash_normal(n, seed, std)
    if dtype == torch.bfloat16:
        ref = synth.bf16_round(ref)
    if dtype == torch.float16:
        ref = ref.astype(np.float16).astype(np.float32)          # IEEE half, round to nearest even
    assert np.array_equal(t.float().cpu().numpy(), ref)
    # offset + add (norm weights)
    _lib.check(lib.atspeed_fill_hash_normal(t.data_ptr(), 1000, seed, float(synth.normal_scale(0.1)), 1.0, code, 0, _st()))
    ref = np.float32(1.0) + synth.hash_normal(1000, seed, 0.1)
    if dtype == torch.bfloat16:
        ref = synth.bf16_round(ref)
    if dtype == torch.float16:
        ref = ref.astype(np.float16).astype(np.float32)
    assert np.array_equal(t[:1000].float().cpu().numpy(), ref)


# ------------------------------------------------------------------ gemm
def _gemm(lib, a, w, epi, dtype, resid=None, n_out=None):
    m, k = a.shape
    n = w.shape[0]
    code = _lib.dtype_code(dtype)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    if epi == _lib.EPI_F32:
        c = torch.zeros(m, n, dtype=torch.float32, device="cuda")
    elif epi == _lib.EPI_SWIGLU:
        c = torch.zeros(m, n // 2, dtype=dtype, device="cuda")
    elif epi == _lib.EPI_RESID:
        c = resid.clone()
    else:
        c = torch.zeros(m, n, dtype=dtype, device="cuda")
    _lib.check(lib.atspeed_gemm(a.data_ptr(), w.data_ptr(), c.data_ptr(), m, n, k, a.stride(0), c.stride(0), code, epi,
                                ws.data_ptr(), ws.numel(), _st()))
    torch.cuda.synchronize()
    return c


SHAPES = [(900, 1000, 512), (1024, 2304, 768), (1543, 300, 1024), (1, 128, 128), (5, 200, 96), (16, 384, 352), (20, 4096, 768), (40, 2304, 768), (61, 768, 3072),
          (100, 512, 4096), (228, 1024, 1024), (228, 4096, 4096), (130, 32256, 128), (121, 32859, 768),
          (257, 640, 1376),
          # one user's wide projections: ring kernel in split-K mode (33-256 tokens, N >= 8192, K % 128 == 0)
          (40, 8192, 512), (100, 12288, 1024), (129, 8448, 256), (228, 12288, 640), (256, 9000, 384),
          # 257-512 tokens: two 256-row token tiles per weight tile in the same mode
          (300, 12288, 640), (400, 8448, 256), (512, 9000, 384),
          # more than two tiles per CU, ragged last tiles
          (4200, 8192, 256), (2100, 16700, 128), (8300, 4100, 384),
          # 16-byte epilogue stores (row stride % 8 == 0) with a last column tile that straddles N: per-wave fallback to element stores
          (4200, 8200, 256)]


@pytest.mark.parametrize("m,n,k", SHAPES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_gemm_store_and_f32(lib, m, n, k, dtype):
    a = _rand((m, k), 1).to(dtype).cuda()
    w = _rand((n, k), 2, 0.05).to(dtype).cuda()
    ref = (a.double().cpu() @ w.double().cpu().T)
    scale = float(ref.abs().max())
    c32 = _gemm(lib, a, w, _lib.EPI_F32, dtype)
    tol = 2e-5 * scale * max(1.0, np.sqrt(k) / 8) if dtype == torch.float32 else 2e-5 * scale * np.sqrt(k)
    np.testing.assert_allclose(c32.double().cpu().numpy(), ref.numpy(), atol=tol, rtol=0)
    c = _gemm(lib, a, w, _lib.EPI_STORE, dtype)
    tol2 = tol if dtype == torch.float32 else 1e-2 * scale
    np.testing.assert_allclose(c.double().cpu().numpy(), ref.numpy(), atol=tol2, rtol=0)


@pytest.mark.parametrize("m,n,k", [(7, 256, 128), (40, 768, 768), (228, 4096, 11008), (228, 4096, 4096), (1100, 768, 1024), (2000, 500, 256), (100, 8192, 512), (6400, 4096, 512), (8300, 4096, 512), (4500, 8192, 256), (6400, 4104, 512),
                                   (50, 8192, 1024), (128, 12288, 768), (129, 8200, 384), (256, 8192, 256), (100, 22016, 512), (60, 32859, 576), (250, 22016, 1024), (100, 12288, 4096), (60, 4096, 11008), (225, 4096, 11008), (128, 12288, 2048),
                                   (320, 4096, 11008), (320, 4096, 4096), (400, 4096, 11008)])      # ... the split-K form of the weight-streaming kernel (qkv, down), and the 257-640-token band (panel split, split-K tail)      # one user's wide projections: the no-split weight-streaming tiles (64 / 128 / 256 token rows)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_gemm_residual(lib, m, n, k, dtype):
    a = _rand((m, k), 3).to(dtype).cuda()
    w = _rand((n, k), 4, 0.03).to(dtype).cuda()
    r = _rand((m, n), 5).to(dtype).cuda()
    ref = r.double().cpu() + a.double().cpu() @ w.double().cpu().T
    c = _gemm(lib, a, w, _lib.EPI_RESID, dtype, resid=r)
    tol = 1e-4 * float(ref.abs().max()) if dtype == torch.float32 else 2e-2 * float(ref.abs().max())
    np.testing.assert_allclose(c.double().cpu().numpy(), ref.numpy(), atol=tol, rtol=0)


@pytest.mark.parametrize("m,ffn,k", [(3, 32, 64), (40, 352, 128), (228, 11008, 4096), (20, 3072, 768), (1300, 1376, 512), (800, 496, 256), (300, 4224, 256), (4200, 4224, 256), (4200, 4240, 256), (60, 4224, 256), (110, 11008, 4096), (40, 11008, 512), (256, 11008, 1024)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_gemm_swiglu(lib, m, ffn, k, dtype):
    from atspeed_amd.model import _interleave_gate_up
    a = _rand((m, k), 6).to(dtype).cuda()
    g = _rand((ffn, k), 7, 0.05).to(dtype).cuda()
    u = _rand((ffn, k), 8, 0.05).to(dtype).cuda()
    wgu = _interleave_gate_up(g, u)
    ad = a.double().cpu()
    ref = torch.nn.functional.silu(ad @ g.double().cpu().T) * (ad @ u.double().cpu().T)
    c = _gemm(lib, a, wgu, _lib.EPI_SWIGLU, dtype)
    tol = 1e-4 * float(ref.abs().max()) if dtype == torch.float32 else 3e-2 * float(ref.abs().max())
    np.testing.assert_allclose(c.double().cpu().numpy(), ref.numpy(), atol=tol, rtol=0)


# ------------------------------------------------------------------ rmsnorm
@pytest.mark.parametrize("hidden", [768, 4096, 8192, 11008, 100])       # vectorised bf16 rows up to 8192, scalar beyond / unaligned
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_rmsnorm(lib, dtype, hidden):
    x = _rand((37, hidden), 9, 2.0).to(dtype).cuda()
    w = (1 + _rand((hidden,), 10, 0.1)).to(dtype).cuda()
    y = torch.empty_like(x)
    code = _lib.dtype_code(dtype)
    _lib.check(lib.atspeed_rmsnorm(x.data_ptr(), w.data_ptr(), y.data_ptr(), 37, hidden, 1e-6, code, _st()))
    xf = x.float().cpu()
    ref = w.float().cpu() * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6))
    np.testing.assert_allclose(y.float().cpu().numpy(), ref.numpy(), atol=1e-5 if dtype == torch.float32 else 4e-2, rtol=0)


# ------------------------------------------------------------------ lse
@pytest.mark.parametrize("rows,V", [(1, 32859), (121, 32859), (40, 33014), (3, 1000), (2, 7)])
def test_lse_rows(lib, rows, V):
    ld = (V + 63) // 64 * 64
    x = torch.full((rows, ld), 1e30)              # padding columns must be ignored
    x[:, :V] = _rand((rows, V), 11, 3.0)
    x[0, 5] = 40.0                                  # one dominant logit
    xd = x.cuda()
    out = torch.empty(rows, device="cuda")
    _lib.check(lib.atspeed_lse_rows(xd.data_ptr(), rows, V, ld, out.data_ptr(), _st()))
    ref = torch.logsumexp(x[:, :V].double(), -1)
    np.testing.assert_allclose(out.double().cpu().numpy(), ref.numpy(), atol=2e-6 * float(ref.abs().max()) + 1e-6, rtol=0)


# ------------------------------------------------------------------ expand + prune
def _dev_fsm(lib, fsm, V):
    rp, tk, nx = (np.ascontiguousarray(x, np.int32) for x in (fsm.row_ptr, fsm.tok, fsm.nxt))
    h = C.c_void_p()
    _lib.check(lib.atspeed_fsm_create(rp.ctypes.data, tk.ctypes.data, nx.ctypes.data, fsm.n_nodes, len(tk), V, C.byref(h)))
    return h


@pytest.mark.parametrize("vocab,rows,k,mode", [(synth.BEAUTY, 1, 20, "pos"), (synth.BEAUTY, 20, 20, "pos"),
                                                (synth.BEAUTY, 40, 40, "pos"), (synth.GAMES, 40, 40, "pos"),
                                                (synth.GAMES, 64, 64, "pos"), (synth.BEAUTY, 40, 40, "trie"),
                                                (synth.BEAUTY, 20, 1, "pos")])
def test_beam_expand_prune_equals_oracle(lib, vocab, rows, k, mode):
    V = vocab.vocab_size
    ld = (V + 63) // 64 * 64
    logits = _rand((rows, V), 21 + rows, 2.5)
    beam_scores = -torch.rand(rows, generator=torch.Generator().manual_seed(rows)) * 5
    if rows > 3:
        beam_scores[2] = float("-inf")             # a dead beam must never be expanded
    prompt = list(synth.synthetic_prompt(16, 4))
    items = synth.synthetic_items(vocab)
    rng = np.random.default_rng(rows * 7 + k)
    if mode == "pos":
        depth = int(rng.integers(0, 4))
        con = PositionSetConstraint(vocab.allowed_tokens(), synth.RESPONSE_SEP)
        suffixes = [[int(items[rng.integers(len(items))][d]) for d in range(depth)] for _ in range(rows)]
    else:
        from atspeed_amd.generation_trie import SuffixTrieConstraint
        con = SuffixTrieConstraint(Trie([[1] + [int(t) for t in it] + [2] for it in items]), synth.RESPONSE_SEP, 1)
        suffixes = [[int(t) for t in items[rng.integers(len(items))][: int(rng.integers(0, 4))]] for _ in range(rows)]
    fsm0 = con.compile(prompt)
    nodes = []
    for s in suffixes:
        nd = fsm0.start
        for t in s:
            nd = fsm0.step(nd, t)
        nodes.append(nd)
    # oracle: sequences prompt + suffix, reference-style callable
    L = max(len(s) for s in suffixes)
    if mode == "pos":
        seqs = torch.tensor([prompt + s for s in suffixes])
        lp = torch.log_softmax(logits, -1)
        masked = R.constrain(seqs, lp, con)
    else:
        lp = torch.log_softmax(logits, -1)
        masked = torch.full_like(lp, float("-inf"))
        for r, s in enumerate(suffixes):
            al = con(0, torch.tensor(prompt + s))
            masked[r, al] = lp[r, al]
    flat = (masked + beam_scores[:, None]).reshape(-1)
    vals, idx = R.topk_desc_stable(flat, k)
    h = _dev_fsm(lib, fsm0, V)
    lg = torch.zeros(rows, ld)
    lg[:, :V] = logits
    lg = lg.cuda()
    lse = torch.empty(rows, device="cuda")
    _lib.check(lib.atspeed_lse_rows(lg.data_ptr(), rows, V, ld, lse.data_ptr(), _st()))
    bs, nd = beam_scores.cuda(), torch.tensor(nodes, dtype=torch.int32).cuda()
    o_s = torch.empty(k, device="cuda")
    o_p, o_t, o_n, o_f = (torch.empty(k, dtype=torch.int32, device="cuda") for _ in range(4))
    _lib.check(lib.atspeed_beam_expand_prune(lg.data_ptr(), ld, lse.data_ptr(), bs.data_ptr(), nd.data_ptr(), rows, h, k,
                                             o_s.data_ptr(), o_p.data_ptr(), o_t.data_ptr(), o_n.data_ptr(), o_f.data_ptr(), _st()))
    torch.cuda.synchronize()
    finite = torch.isfinite(vals)
    nfin = int(finite.sum())
    assert o_f.cpu()[:nfin].tolist() == idx[:nfin].tolist()              # bit-exact candidate ids, in order
    assert (o_f.cpu()[nfin:] == -1).all()
    np.testing.assert_allclose(o_s.cpu()[:nfin].numpy(), vals[:nfin].numpy(), atol=1e-4, rtol=0)
    assert o_p.cpu()[:nfin].tolist() == (idx[:nfin] // V).tolist()
    assert o_t.cpu()[:nfin].tolist() == (idx[:nfin] % V).tolist()
    for j in range(nfin):
        assert int(o_n[j]) == fsm0.step(nodes[int(o_p[j])], int(o_t[j]))
    lib.atspeed_fsm_destroy(h)


def test_constraint_dead_end_raises(lib):
    V = synth.TINY.vocab_size
    con = PositionSetConstraint(synth.TINY.allowed_tokens(), synth.RESPONSE_SEP)
    fsm = con.compile(list(synth.synthetic_prompt(8, 1)))
    h = _dev_fsm(lib, fsm, V)
    # node 5 (after EOS) allows nothing: HF raises ValueError; here the standalone op just returns no candidates
    ld = (V + 63) // 64 * 64
    lg = torch.zeros(1, ld, device="cuda")
    lse = torch.zeros(1, device="cuda")
    bs = torch.zeros(1, device="cuda")
    nd = torch.tensor([5], dtype=torch.int32, device="cuda")
    o_s = torch.empty(4, device="cuda")
    o = [torch.empty(4, dtype=torch.int32, device="cuda") for _ in range(4)]
    _lib.check(lib.atspeed_beam_expand_prune(lg.data_ptr(), ld, lse.data_ptr(), bs.data_ptr(), nd.data_ptr(), 1, h, 4,
                                             o_s.data_ptr(), *[x.data_ptr() for x in o], _st()))
    torch.cuda.synchronize()
    assert (o[3].cpu() == -1).all()
    lib.atspeed_fsm_destroy(h)


# ------------------------------------------------------------------ accept
@pytest.mark.parametrize("k,dk,kind", [(20, 40, "all"), (20, 40, "miss"), (20, 20, "all"), (1, 1, "all"), (10, 64, "all"),
                                        (20, 40, "invalid")])
def test_accept_equals_reference_logic(lib, k, dk, kind):
    rng = np.random.default_rng(k * 100 + dk)
    draft = rng.choice(10 ** 6, size=dk, replace=False).astype(np.int32)
    pos = rng.choice(dk, size=k, replace=False)
    target = draft[pos].copy()
    if kind == "miss":
        target[k // 2] = 10 ** 6 + 5
    if kind == "invalid":
        target[0] = -1
    tscore = -np.sort(rng.random(k)).astype(np.float32)
    # beamSD.py:371-380 in plain Python
    hit_ref = [i for i, d in enumerate(draft.tolist()) if d in set(target.tolist())]
    found = [draft.tolist().index(y) for y in target.tolist() if y in draft.tolist()]
    order = np.argsort(np.array(found), kind="stable")
    accept_ref = len(hit_ref) == k
    t, ts, d = torch.from_numpy(target).cuda(), torch.from_numpy(tscore).cuda(), torch.from_numpy(draft).cuda()
    hit = torch.empty(k, dtype=torch.int32, device="cuda")
    sbh = torch.empty(k, device="cuda")
    acc = torch.empty(1, dtype=torch.int32, device="cuda")
    _lib.check(lib.atspeed_accept(t.data_ptr(), ts.data_ptr(), k, d.data_ptr(), dk, hit.data_ptr(), sbh.data_ptr(), acc.data_ptr(), _st()))
    torch.cuda.synchronize()
    assert bool(acc.item()) == accept_ref
    if accept_ref:
        assert hit.cpu().tolist() == hit_ref
        assert np.array_equal(sbh.cpu().numpy(), tscore[order])


# ------------------------------------------------------------------ attention
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T", [23, 150, 230])       # 150 rows take the 8-wave (128-row tile) kernel, 230 the 16-wave (256-row) one
@pytest.mark.parametrize("heads,dh", [(4, 32), (12, 64), (32, 128)])
def test_tree_attention(lib, dtype, heads, dh, T):
    from atspeed_amd.model import vis_bits_from_bool
    S, max_slots = 150, 256
    H = heads * dh
    q = _rand((T, 3 * H), 31).to(dtype).cuda()
    kc = _rand((max_slots, H), 32).to(dtype).cuda()
    vc = _rand((max_slots, H), 33).to(dtype).cuda()
    g = torch.Generator().manual_seed(5)
    vis = torch.rand(T, S, generator=g) < 0.3
    vis[:, 0] = True
    vis[3] = False
    vis[3, 149] = True           # a row with a single visible slot at the very end
    bits = vis_bits_from_bool(vis, max_slots).cuda()
    out = torch.zeros(T, H, dtype=dtype, device="cuda")
    code = _lib.dtype_code(dtype)
    _lib.check(lib.atspeed_tree_attention(q.data_ptr(), 3 * H, kc.data_ptr(), vc.data_ptr(), bits.data_ptr(), max_slots // 64,
                                          out.data_ptr(), T, S, heads, dh, code, _st()))
    torch.cuda.synchronize()
    qf = q.float().cpu()[:, :H].view(T, heads, dh)
    kf = kc.float().cpu()[:S].view(S, heads, dh)
    vf = vc.float().cpu()[:S].view(S, heads, dh)
    sc = torch.einsum("thd,shd->hts", qf, kf) / np.sqrt(dh)
    sc = sc.masked_fill(~vis[None], float("-inf"))
    ref = torch.einsum("hts,shd->thd", torch.softmax(sc, -1), vf).reshape(T, H)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref.numpy(), atol=2e-5 if dtype == torch.float32 else 3e-2, rtol=0)


@pytest.mark.parametrize("rows_per_wave", [16, 32])
@pytest.mark.parametrize("qtile", [64, 128, 256])
@pytest.mark.parametrize("heads,dh,T,S", [(12, 64, 200, 330), (32, 128, 300, 470), (8, 128, 37, 64), (4, 64, 129, 1), (32, 128, 600, 200)])
def test_tree_attention_every_tiling(lib, heads, dh, T, S, qtile, rows_per_wave):
    """The MFMA kernels (16 rows per wave: a four-tile LDS-DMA ring up to 256 workgroups -- one user's forwards --, register-staged tiles
    above that (the 600-row case) and for the 256-row tile; 32 rows per wave, LDS-DMA double buffer) at every query-tile height: ragged last
    tiles (T, S not multiples of 64), a single-slot cache (fewer tiles than ring stages), 8 tiles through a 4-stage ring, rows that see
    one far slot, rows that see nothing in whole 64-slot tiles."""
    from atspeed_amd.model import vis_bits_from_bool
    max_slots = 512
    H = heads * dh
    q = _rand((T, 3 * H), 41).to(torch.bfloat16).cuda()
    kc = _rand((max_slots, H), 42).to(torch.bfloat16).cuda()
    vc = _rand((max_slots, H), 43).to(torch.bfloat16).cuda()
    g = torch.Generator().manual_seed(9)
    vis = torch.rand(T, S, generator=g) < 0.25
    vis[:, 0] = True
    if S > 100:
        vis[5] = False; vis[5, S - 1] = True        # one visible slot, in the last (ragged) tile
        vis[7, 64:] = False                         # nothing beyond the first tile
        vis[T - 1, : S - 3] = False; vis[T - 1, S - 3:] = True
    bits = vis_bits_from_bool(vis, max_slots).cuda()
    out = torch.full((T, H), 7.0, dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.atspeed_tree_attention_tiled(q.data_ptr(), 3 * H, kc.data_ptr(), vc.data_ptr(), bits.data_ptr(), max_slots // 64,
                                                out.data_ptr(), T, S, heads, dh, _lib.ATSPEED_BF16, qtile, rows_per_wave, _st()))
    torch.cuda.synchronize()
    qf = q.float().cpu()[:, :H].view(T, heads, dh)
    kf = kc.float().cpu()[:S].view(S, heads, dh)
    vf = vc.float().cpu()[:S].view(S, heads, dh)
    sc = torch.einsum("thd,shd->hts", qf, kf) / np.sqrt(dh)
    sc = sc.masked_fill(~vis[None], float("-inf"))
    ref = torch.einsum("hts,shd->thd", torch.softmax(sc, -1), vf).reshape(T, H)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref.numpy(), atol=3e-2, rtol=0)


def test_tree_attention_tiled_rejects_bad_tiling(lib):
    z = torch.zeros(64, 3 * 64, dtype=torch.bfloat16, device="cuda")
    bits = torch.zeros(64, 1, dtype=torch.int64, device="cuda")
    for qtile, rpw in ((100, 0), (64, 8)):
        with pytest.raises(_lib.AtSpeedError):
            _lib.check(lib.atspeed_tree_attention_tiled(z.data_ptr(), 192, z.data_ptr(), z.data_ptr(), bits.data_ptr(), 1, z.data_ptr(), 64, 64, 1, 64,
                                                        _lib.ATSPEED_BF16, qtile, rpw, _st()))


# ------------------------------------------------------------------ fp8 (BASELINE config 5)
def _fp8_to_float(q_u8: torch.Tensor) -> torch.Tensor:
    return q_u8.cpu().view(torch.float8_e4m3fn).to(torch.float32)


@pytest.mark.parametrize("rows,cols", [(37, 1152), (64, 4096), (51, 11008), (6, 64), (9, 12352)])
def test_quant_rows_fp8_packed_equals_the_row_major_kernel_bit_for_bit(lib, rows, cols):
    """the row-pair kernel of the packed layout (whole 128-byte lines, a pair in registers; 12352 columns: the one-row fallback) against
    atspeed_quant_rows_fp8 on the same values; odd row counts leave the pad row of the last pair alone"""
    x = (_rand((rows, cols), 43, 3.0) * torch.linspace(0.01, 30, rows)[:, None]).to(torch.bfloat16).cuda()
    x[min(5, rows - 1)] = 0
    q0 = torch.empty(rows, cols, dtype=torch.uint8, device="cuda")
    s0 = torch.empty(rows, dtype=torch.float32, device="cuda")
    _lib.check(lib.atspeed_quant_rows_fp8(x.data_ptr(), rows, cols, q0.data_ptr(), s0.data_ptr(), _st()))
    re = (rows + 1) // 2 * 2
    xp = torch.zeros(re, cols, dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.atspeed_pack_rows(x.data_ptr(), xp.data_ptr(), rows, cols * 2, _st()))
    qp = torch.full((re, cols), 0xAB, dtype=torch.uint8, device="cuda")
    s1 = torch.full((rows + 1,), -7.0, dtype=torch.float32, device="cuda")
    _lib.check(lib.atspeed_quant_rows_fp8_packed(xp.data_ptr(), rows, cols, qp.data_ptr(), s1.data_ptr(), _st()))
    q1 = torch.empty(rows, cols, dtype=torch.uint8, device="cuda")
    _lib.check(lib.atspeed_unpack_rows(qp.data_ptr(), q1.data_ptr(), rows, cols, _st()))
    torch.cuda.synchronize()
    assert torch.equal(q0, q1)
    assert torch.equal(s0, s1[:rows]) and float(s1[rows]) == -7.0            # no scale written for the pad row


def test_quant_rows_fp8(lib):
    rows, cols = 37, 1152
    x = (_rand((rows, cols), 41, 3.0) * torch.linspace(0.01, 30, rows)[:, None]).to(torch.bfloat16).cuda()
    x[5] = 0                                                     # an all-zero row must not divide by zero
    q = torch.empty(rows, cols, dtype=torch.uint8, device="cuda")
    sc = torch.empty(rows, dtype=torch.float32, device="cuda")
    _lib.check(lib.atspeed_quant_rows_fp8(x.data_ptr(), rows, cols, q.data_ptr(), sc.data_ptr(), _st()))
    torch.cuda.synchronize()
    xf = x.float().cpu()
    amax = xf.abs().amax(1)
    exp_scale = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    np.testing.assert_allclose(sc.cpu().numpy(), exp_scale.numpy(), rtol=1e-6)
    deq = _fp8_to_float(q) * sc.cpu()[:, None]
    # e4m3 has 3 mantissa bits: relative error <= 2^-4 of the value (plus the subnormal floor of the row)
    err = (deq - xf).abs()
    bound = xf.abs() * 2 ** -4 + exp_scale[:, None] * 2 ** -9 + 1e-30
    assert bool((err <= bound * 1.001).all())
    ref_codes = (xf / sc.cpu()[:, None]).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    assert (ref_codes == q.cpu()).float().mean() > 0.999          # same rounding as torch's e4m3 conversion


@pytest.mark.parametrize("hidden", [768, 4096, 8192])
def test_rmsnorm_quant_fp8_equals_norm_then_quant(lib, hidden):
    rows = 37
    x = _rand((rows, hidden), 41, 2.0).to(torch.bfloat16).cuda()
    w = (1 + _rand((hidden,), 42, 0.1)).to(torch.bfloat16).cuda()
    y = torch.empty_like(x); y2 = torch.empty_like(x)
    q = torch.empty(rows, hidden, dtype=torch.uint8, device="cuda"); q2 = torch.empty_like(q)
    sc = torch.empty(rows, dtype=torch.float32, device="cuda"); sc2 = torch.empty_like(sc)
    _lib.check(lib.atspeed_rmsnorm(x.data_ptr(), w.data_ptr(), y.data_ptr(), rows, hidden, 1e-6, _lib.ATSPEED_BF16, _st()))
    _lib.check(lib.atspeed_quant_rows_fp8(y.data_ptr(), rows, hidden, q.data_ptr(), sc.data_ptr(), _st()))
    _lib.check(lib.atspeed_rmsnorm_quant_fp8(x.data_ptr(), w.data_ptr(), y2.data_ptr(), q2.data_ptr(), sc2.data_ptr(), rows, hidden, 1e-6, _st()))
    torch.cuda.synchronize()
    assert torch.equal(y.view(torch.int16), y2.view(torch.int16)) and torch.equal(q, q2) and torch.equal(sc, sc2)
    q3 = torch.empty_like(q); sc3 = torch.empty_like(sc)                      # y = NULL: only the quantised row
    _lib.check(lib.atspeed_rmsnorm_quant_fp8(x.data_ptr(), w.data_ptr(), None, q3.data_ptr(), sc3.data_ptr(), rows, hidden, 1e-6, _st()))
    torch.cuda.synchronize()
    assert torch.equal(q, q3) and torch.equal(sc, sc3)


def test_gemm_fp8_refuses_an_operand_beyond_the_kernels_32_bit_row_offsets(lib):
    """the LDS-DMA kernels address an operand as a 64-bit base + a 32-bit per-lane byte offset: an operand of 4 GB or more is refused
    (ATSPEED_ERR_CAPACITY) before anything is launched -- the pointers here are never dereferenced"""
    x = torch.zeros(64, dtype=torch.uint8, device="cuda")
    s = torch.zeros(64, dtype=torch.float32, device="cuda")
    rc = lib.atspeed_gemm_fp8(x.data_ptr(), s.data_ptr(), x.data_ptr(), s.data_ptr(), x.data_ptr(), 512, 70000, 65536, 70000, 0, None, 0, _st())
    assert rc == _lib.ERR_CAPACITY and b"32-bit" in lib.atspeed_last_error()


@pytest.mark.parametrize("m,n,k,epi", [(1024, 2304, 768, 0), (640, 12288, 512, 0), (1300, 1024, 1280, 2), (900, 2752, 512, 3), (1543, 1000, 256, 1),
                                      (4200, 8192, 256, 0), (8300, 4096, 512, 2), (4200, 8448, 256, 3), (4480, 3072, 1024, 0),      # more than two tiles per CU
                                      # the Llama-7B projections at their real K (config 5): qkv, o_proj, gate_up, down
                                      (1100, 12288, 4096, 0), (1100, 4096, 4096, 2), (700, 22016, 4096, 3), (1100, 4096, 11008, 2)])
def test_gemm_fp8(lib, m, n, k, epi):
    x = _rand((m, k), 51, 1.5).to(torch.bfloat16).cuda()
    w = _rand((n, k), 52, 0.05).to(torch.bfloat16).cuda()
    xq = torch.empty(m, k, dtype=torch.uint8, device="cuda"); sx = torch.empty(m, device="cuda")
    wq = torch.empty(n, k, dtype=torch.uint8, device="cuda"); sw = torch.empty(n, device="cuda")
    _lib.check(lib.atspeed_quant_rows_fp8(x.data_ptr(), m, k, xq.data_ptr(), sx.data_ptr(), _st()))
    _lib.check(lib.atspeed_quant_rows_fp8(w.data_ptr(), n, k, wq.data_ptr(), sw.data_ptr(), _st()))
    # exact reference of what the kernel must compute: dequantised operands, fp64 accumulate, per-row scales
    ref = (_fp8_to_float(xq).double() @ _fp8_to_float(wq).double().T) * sx.cpu().double()[:, None] * sw.cpu().double()[None, :]
    if epi == _lib.EPI_F32:
        ldc = (n + 63) // 64 * 64
        c = torch.zeros(m, ldc, dtype=torch.float32, device="cuda")
    elif epi == _lib.EPI_SWIGLU:
        ldc = n // 2
        c = torch.zeros(m, ldc, dtype=torch.bfloat16, device="cuda")
        g = ref.view(m, n // 32, 2, 16)
        ref = (torch.nn.functional.silu(g[:, :, 0]) * g[:, :, 1]).reshape(m, n // 2)
    else:
        ldc = n
        c = (_rand((m, n), 53).to(torch.bfloat16).cuda() if epi == _lib.EPI_RESID else torch.zeros(m, n, dtype=torch.bfloat16, device="cuda"))
        if epi == _lib.EPI_RESID:
            ref = ref + c.double().cpu()
    _lib.check(lib.atspeed_gemm_fp8(xq.data_ptr(), sx.data_ptr(), wq.data_ptr(), sw.data_ptr(), c.data_ptr(), m, n, k, ldc, epi, None, 0, _st()))
    torch.cuda.synchronize()
    out = c.double().cpu()[:, : ref.shape[1]]
    scale = float(ref.abs().max())
    tol = 2e-5 * scale * np.sqrt(k) if epi == _lib.EPI_F32 else 2e-2 * scale
    np.testing.assert_allclose(out.numpy(), ref.numpy(), atol=tol, rtol=0)


# 70001: more than 256 tiles of 256 columns (a 128k-token target would have 500): the tile mask is sized by the vocabulary
@pytest.mark.parametrize("rows,vocab,hidden", [(700, 32859, 512), (1300, 33014, 256), (2100, 5000, 768), (90, 32859, 256), (300, 70001, 256), (700, 32859, 2048), (1300, 33014, 1024)])      # the last two: K deep enough for the stream-K tail
def test_lmhead_lse_fused_epilogue(lib, rows, vocab, hidden):
    """lm_head + full-vocabulary normaliser in one kernel (beamSD.py:58,285): lse equals logsumexp of the fp32 product over ALL columns
    (tail tile of a vocabulary that is no multiple of 256 included), the logit tiles of the automaton's tokens are the plain GEMM's bit for
    bit, every other tile is left untouched (never written: that is the HBM traffic the fusion removes).  90 rows: the small path
    (plain GEMM + streaming LSE), same answers, everything written."""
    x = _rand((rows, hidden), 61, 1.0).to(torch.bfloat16).cuda()
    w = _rand((vocab, hidden), 62, 0.08).to(torch.bfloat16).cuda()
    ld = (vocab + 63) // 64 * 64
    ref = x.float().cpu().double() @ w.float().cpu().double().T
    ref_lse = torch.logsumexp(ref, dim=1)
    plain = torch.zeros(rows, ld, dtype=torch.float32, device="cuda")
    ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    _lib.check(lib.atspeed_gemm(x.data_ptr(), w.data_ptr(), plain.data_ptr(), rows, vocab, hidden, hidden, ld, _lib.ATSPEED_BF16, _lib.EPI_F32,
                                ws.data_ptr(), ws.numel(), _st()))
    # automaton over a few token ids: tiles {0, last two full ones, tail}
    toks = sorted({2, 300 % vocab, vocab - 600, vocab - 300, vocab - 1} | ({65536 + 5, 66000} if vocab > 66000 else set()))
    row_ptr = np.array([0, len(toks), len(toks)], np.int32); tok = np.array(toks, np.int32); nxt = np.ones(len(toks), np.int32)
    fsm = C.c_void_p()
    _lib.check(lib.atspeed_fsm_create(row_ptr.ctypes.data, tok.ctypes.data, nxt.ctypes.data, 2, len(toks), vocab, C.byref(fsm)))
    try:
        for use_fsm in (False, True):
            SENT = -12345.0
            logits = torch.full((rows, ld), SENT, dtype=torch.float32, device="cuda")
            lse = torch.empty(rows, dtype=torch.float32, device="cuda")
            fused = C.c_int32(-1)
            _lib.check(lib.atspeed_lmhead_lse(x.data_ptr(), w.data_ptr(), logits.data_ptr(), lse.data_ptr(), rows, vocab, hidden, ld,
                                              fsm if use_fsm else None, ws.data_ptr(), ws.numel(), C.byref(fused), _st()))
            torch.cuda.synchronize()
            assert fused.value == (1 if rows >= 257 else 0)
            np.testing.assert_allclose(lse.cpu().double().numpy(), ref_lse.numpy(), atol=2e-4, rtol=0)
            stored = {t // 256 for t in toks} if (use_fsm and fused.value) else set(range((vocab + 255) // 256))
            for t in range((vocab + 255) // 256):
                lo, hi = t * 256, min(vocab, t * 256 + 256)
                got = logits[:, lo:hi]
                if t in stored:
                    assert torch.equal(got, plain[:, lo:hi]), f"tile {t} differs from the plain GEMM"
                else:
                    assert bool((got == SENT).all()), f"tile {t} was written although no token of the automaton lies in it"
    finally:
        lib.atspeed_fsm_destroy(fsm)


# ------------------------------------------------------------------ packed operand layout (what the bf16 / fp8 engine feeds its GEMMs)
def _pack(lib, t):
    rows, cols = t.shape
    out = torch.empty((rows + 1) // 2 * 2, cols, dtype=t.dtype, device="cuda")
    _lib.check(lib.atspeed_pack_rows(t.data_ptr(), out.data_ptr(), rows, cols * t.element_size(), _st()))
    return out


def _unpack(lib, t, rows):
    out = torch.empty(rows, t.shape[1], dtype=t.dtype, device="cuda")
    _lib.check(lib.atspeed_unpack_rows(t.data_ptr(), out.data_ptr(), rows, t.shape[1] * t.element_size(), _st()))
    return out


def test_pack_rows_layout_and_round_trip(lib):
    """byte b of row r -> ((r >> 1) * (row_bytes / 64) + (b >> 6)) * 128 + (r & 1) * 64 + (b & 63); odd row counts get a zero pad row."""
    for rows, cols, dt in ((7, 96, torch.bfloat16), (64, 4096, torch.bfloat16), (33, 256, torch.uint8), (2, 32, torch.bfloat16)):
        src = (torch.arange(rows * cols, dtype=torch.int32) % 251).reshape(rows, cols).to(dt).cuda()
        pk = _pack(lib, src)
        torch.cuda.synchronize()
        rb = cols * src.element_size()
        flat_src = src.contiguous().view(torch.uint8).cpu().numpy().reshape(rows, rb)
        flat_pk = pk.view(torch.uint8).cpu().numpy().reshape(-1)
        r, b = np.meshgrid(np.arange(rows), np.arange(rb), indexing="ij")
        off = ((r >> 1) * (rb // 64) + (b >> 6)) * 128 + (r & 1) * 64 + (b & 63)
        assert np.array_equal(flat_pk[off], flat_src)
        if rows % 2:
            rr = np.full(rb, rows); bb = np.arange(rb)
            assert not flat_pk[((rr >> 1) * (rb // 64) + (bb >> 6)) * 128 + (rr & 1) * 64 + (bb & 63)].any()
        assert torch.equal(_unpack(lib, pk, rows), src)
    with pytest.raises(_lib.AtSpeedError):
        x = torch.zeros(4, 40, dtype=torch.bfloat16, device="cuda")          # 80-byte rows: not a multiple of 64
        lib_out = torch.zeros(4, 40, dtype=torch.bfloat16, device="cuda")
        _lib.check(lib.atspeed_pack_rows(x.data_ptr(), lib_out.data_ptr(), 4, 80, _st()))


@pytest.mark.parametrize("m,n,k,epi", [(20, 768, 256, 0), (100, 12288, 512, 0), (228, 4096, 4096, 2), (228, 22016, 512, 3), (121, 32859, 256, 1), (50, 12288, 512, 0), (50, 8448, 256, 3), (121, 32859, 512, 1), (110, 22016, 512, 3), (64, 8448, 1024, 3), (128, 8192, 4096, 0), (33, 16384, 576, 0), (225, 22016, 512, 3), (256, 12288, 1024, 0), (129, 32859, 512, 1), (64, 22016, 1024, 3), (40, 32859, 576, 1), (100, 22016, 512, 0), (200, 22000, 512, 0), (128, 28672, 512, 1),
                                      (1300, 1024, 1280, 2), (900, 2752, 512, 3), (4200, 8192, 256, 0), (1543, 1001, 256, 1), (777, 4096, 11008, 2),
                                      # the 257-640-token band at the Llama-7B shapes: panel form, its split form, split-K tail, plain ring, as the default dispatch picks
                                      (320, 22016, 4096, 3), (320, 4096, 11008, 2), (320, 4096, 4096, 2), (400, 12288, 4096, 0), (640, 22016, 4096, 3)])
def test_gemm_packed_equals_row_major_bit_for_bit(lib, m, n, k, epi):
    """Every bf16 GEMM path (LDS-tiled, its split-K + reduce, the split-K ring, the 256-wide ring) on packed operands: the same products
    in the same order as on row-major operands, so the outputs are IDENTICAL (the SwiGLU output comes back packed)."""
    a = _rand((m, k), 71, 1.0).to(torch.bfloat16).cuda()
    w = _rand((n, k), 72, 0.05).to(torch.bfloat16).cuda()
    ws = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    if epi == _lib.EPI_F32:
        ldc = (n + 63) // 64 * 64; mk = lambda: torch.zeros(m, ldc, dtype=torch.float32, device="cuda")
    elif epi == _lib.EPI_SWIGLU:
        ldc = n // 2; mk = lambda: torch.zeros((m + 1) // 2 * 2, ldc, dtype=torch.bfloat16, device="cuda")
    else:
        ldc = n; base = _rand((m, n), 73).to(torch.bfloat16).cuda(); mk = lambda: (base.clone() if epi == _lib.EPI_RESID else torch.zeros_like(base))
    c0, c1 = mk(), mk()
    _lib.check(lib.atspeed_gemm(a.data_ptr(), w.data_ptr(), c0.data_ptr(), m, n, k, k, ldc, _lib.ATSPEED_BF16, epi, ws.data_ptr(), ws.numel(), _st()))
    ap, wp = _pack(lib, a), _pack(lib, w)
    _lib.check(lib.atspeed_gemm_packed(ap.data_ptr(), wp.data_ptr(), c1.data_ptr(), m, n, k, ldc, epi, ws.data_ptr(), ws.numel(), _st()))
    torch.cuda.synchronize()
    if epi == _lib.EPI_SWIGLU:
        assert torch.equal(_unpack(lib, c1, m), c0[:m])
    else:
        assert torch.equal(c1[:, :n], c0[:, :n])


@pytest.mark.parametrize("m,n,k,epi", [(1024, 2304, 768, 0), (1300, 1024, 1280, 2), (900, 2752, 512, 3), (4200, 8448, 256, 3), (8300, 4096, 512, 2), (4480, 3072, 1024, 0),
                                      (1100, 12288, 4096, 0), (700, 22016, 4096, 3), (1100, 4096, 11008, 2)])      # Llama-7B shapes: K = 4096 and K = 11008
def test_gemm_fp8_packed_equals_row_major_bit_for_bit(lib, m, n, k, epi):
    x = _rand((m, k), 51, 1.5).to(torch.bfloat16).cuda()
    w = _rand((n, k), 52, 0.05).to(torch.bfloat16).cuda()
    xq = torch.empty(m, k, dtype=torch.uint8, device="cuda"); sx = torch.empty(m, device="cuda")
    wq = torch.empty(n, k, dtype=torch.uint8, device="cuda"); sw = torch.empty(n, device="cuda")
    _lib.check(lib.atspeed_quant_rows_fp8(x.data_ptr(), m, k, xq.data_ptr(), sx.data_ptr(), _st()))
    _lib.check(lib.atspeed_quant_rows_fp8(w.data_ptr(), n, k, wq.data_ptr(), sw.data_ptr(), _st()))
    ldc = n // 2 if epi == _lib.EPI_SWIGLU else n
    base = _rand((m, ldc), 53).to(torch.bfloat16).cuda()
    mk = lambda: torch.cat((base, torch.zeros(m % 2, ldc, dtype=torch.bfloat16, device="cuda"))) if epi != _lib.EPI_SWIGLU else torch.zeros((m + 1) // 2 * 2, ldc, dtype=torch.bfloat16, device="cuda")
    c0, c1 = mk(), mk()
    _lib.check(lib.atspeed_gemm_fp8(xq.data_ptr(), sx.data_ptr(), wq.data_ptr(), sw.data_ptr(), c0.data_ptr(), m, n, k, ldc, epi, None, 0, _st()))
    xp, wp = _pack(lib, xq), _pack(lib, wq)
    _lib.check(lib.atspeed_gemm_fp8_packed(xp.data_ptr(), sx.data_ptr(), wp.data_ptr(), sw.data_ptr(), c1.data_ptr(), m, n, k, ldc, epi, None, 0, _st()))
    torch.cuda.synchronize()
    if epi == _lib.EPI_SWIGLU:
        assert torch.equal(_unpack(lib, c1, m), c0[:m])
    else:
        assert torch.equal(c1[:m], c0[:m])


def test_measured_peak_probes_are_plausible(lib):
    """bench.py's measured peaks: the probes run, synchronise and return numbers between a loose floor and the
    nominal peaks of /opt/skills/guides/MI355X_MICROARCH.md (2.5 PF dense bf16, 8 TB/s)."""
    tf, gbs = C.c_double(), C.c_double()
    scratch = torch.empty(4 << 20, dtype=torch.uint8, device="cuda")
    buf = torch.ones(1 << 30, dtype=torch.uint8, device="cuda")
    _lib.check(lib.atspeed_probe_mfma_bf16(2000, scratch.data_ptr(), scratch.numel(), _lib.stream_ptr(), C.byref(tf)))
    _lib.check(lib.atspeed_probe_hbm_read(buf.data_ptr(), buf.numel(), 4, scratch.data_ptr(), _lib.stream_ptr(), C.byref(gbs)))
    assert 800.0 < tf.value < 2600.0, tf.value
    assert 2000.0 < gbs.value < 8200.0, gbs.value
    with pytest.raises(_lib.AtSpeedError):
        _lib.check(lib.atspeed_probe_mfma_bf16(10, scratch.data_ptr(), 1024, _lib.stream_ptr(), C.byref(tf)))


# ------------------------------------------------------------------ mask-free search (prefix_allowed_tokens_fn=None) and host-side processors
@pytest.mark.parametrize("rows,vocab,k", [(1, 32859, 20), (40, 32859, 40), (20, 33014, 20), (64, 70001, 64), (7, 300, 5), (3, 16384 + 5, 1)])
def test_row_topk_and_free_expand_equal_torch(lib, rows, vocab, k):
    """`atspeed_row_topk`: the k best columns of every row, (value desc, column asc), -inf never; `atspeed_beam_expand_prune_free`: the k
    best (row, token) pairs of log-softmax + beam score over ALL columns (beamSD.py:58,69-78 with an empty processor list) -- against
    torch's stable sort.  Vocabularies above one 16384-column chunk, above 65536, a dead beam, rows with -inf entries."""
    ld = (vocab + 63) // 64 * 64
    logits = _rand((rows, vocab), 300 + rows, 3.0)
    logits[0, : min(vocab, 50)] = float("-inf")                      # masked entries (what a logits processor leaves behind)
    if rows > 2:
        logits[2, 5:] = float("-inf")                                # fewer finite entries than k
    lg = torch.full((rows, ld), 7.0)
    lg[:, :vocab] = logits
    lg = lg.cuda()
    out = torch.empty(rows, _lib.MAX_BEAMS, dtype=torch.int32, device="cuda")
    _lib.check(lib.atspeed_row_topk(lg.data_ptr(), rows, vocab, ld, k, out.data_ptr(), _st()))
    got = out.cpu()
    for r in range(rows):
        srt = torch.sort(logits[r], descending=True, stable=True)
        want = [int(i) for v, i in zip(srt.values[:k], srt.indices[:k]) if v > float("-inf")]
        assert got[r, : len(want)].tolist() == want, r
        assert bool((got[r, len(want):] == -1).all())
    beam = -torch.rand(rows, generator=torch.Generator().manual_seed(rows)) * 5
    if rows > 3:
        beam[3] = float("-inf")                                      # a dead beam is never expanded
    lse = torch.logsumexp(logits.double(), -1).float()
    flat = ((logits - lse[:, None]) + beam[:, None]).reshape(-1)
    vals, idx = R.topk_desc_stable(flat, k)
    keep = vals > float("-inf")
    vals, idx = vals[keep], idx[keep]
    ws = torch.empty(rows * _lib.MAX_BEAMS, dtype=torch.int32, device="cuda")
    o_s = torch.empty(k, dtype=torch.float32, device="cuda")
    o_p, o_t, o_f = (torch.empty(k, dtype=torch.int32, device="cuda") for _ in range(3))
    lse_d, beam_d = lse.cuda(), beam.cuda()                         # named: a temporary's memory is gone before the kernel runs
    _lib.check(lib.atspeed_beam_expand_prune_free(lg.data_ptr(), ld, lse_d.data_ptr(), beam_d.data_ptr(), rows, vocab, k, ws.data_ptr(),
                                                  o_s.data_ptr(), o_p.data_ptr(), o_t.data_ptr(), o_f.data_ptr(), _st()))
    torch.cuda.synchronize()
    f = o_f.cpu()
    n = int((f >= 0).sum())
    assert n == len(idx) and bool((f[n:] == -1).all())
    assert f[:n].tolist() == idx.tolist()                            # flat ids, in order
    assert o_p.cpu()[:n].tolist() == (idx // vocab).tolist() and o_t.cpu()[:n].tolist() == (idx % vocab).tolist()
    np.testing.assert_allclose(o_s.cpu()[:n].numpy(), vals.numpy(), atol=1e-5, rtol=0)


def test_log_softmax_rows_and_assemble_sequences(lib):
    rows, vocab = 37, 32859
    ld = (vocab + 63) // 64 * 64
    lg = _rand((rows, ld), 17, 2.0).cuda()
    lse = torch.empty(rows, dtype=torch.float32, device="cuda")
    _lib.check(lib.atspeed_lse_rows(lg.data_ptr(), rows, vocab, ld, lse.data_ptr(), _st()))
    out = torch.empty(rows, vocab, dtype=torch.float32, device="cuda")
    _lib.check(lib.atspeed_log_softmax_rows(lg.data_ptr(), ld, lse.data_ptr(), rows, vocab, out.data_ptr(), vocab, _st()))
    assert torch.equal(out, lg[:, :vocab] - lse[:, None])
    np.testing.assert_allclose(out.cpu().numpy(), torch.log_softmax(lg[:, :vocab].double().cpu(), -1).numpy(), atol=2e-5, rtol=0)
    # beam_sequence tensors of a batch in one launch: prompt ++ suffix per beam, ragged prompts
    n, k, L = 9, 20, 4
    g = torch.Generator().manual_seed(3)
    lens = [int(x) for x in torch.randint(1, 200, (n,), generator=g)]
    prompts = [torch.randint(0, 32000, (p,), generator=g, dtype=torch.int32) for p in lens]
    toks = torch.randint(32000, 32859, (n, k, L), generator=g, dtype=torch.int32)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    flat = torch.cat(prompts).cuda()
    res = torch.full((k * (int(off[-1]) + n * L),), -1, dtype=torch.int64, device="cuda")
    _lib.check(lib.atspeed_assemble_sequences(flat.data_ptr(), off.ctypes.data, toks.cuda().data_ptr(), n, k, L, res.data_ptr(), _st()))
    res = res.cpu()
    for u in range(n):
        o0 = k * (int(off[u]) + u * L)
        got = res[o0: o0 + k * (lens[u] + L)].view(k, lens[u] + L)
        want = torch.cat((prompts[u].long()[None].expand(k, -1), toks[u].long()), 1)
        assert torch.equal(got, want), u


# ------------------------------------------------------------------ stream-K tail of the ring GEMM (4-64 users per lock-step batch)
PATH_RING, PATH_RING_SK, PATH_WDMA, PATH_WDMA_SPLIT, PATH_RING_SPLIT, PATH_TILED, PATH_FP8_RING, PATH_FP8_WDMA, PATH_FP8_WDMA_SPLIT, PATH_PANEL, PATH_PANEL_SPLIT = range(11)


def _path_counters(lib, reset=False):
    import ctypes as C
    out = (C.c_int64 * 16)()
    lib.atspeed_gemm_path_counters(out, 16, 1 if reset else 0)
    return list(out)


SK_SHAPES = [(900, 4096, 4096, 2), (900, 4096, 11008, 2), (1800, 12288, 4096, 0), (912, 22016, 4096, 3), (400, 4096, 4096, 2), (912, 12288, 4096, 0),
             (1600, 22016, 4096, 3), (640, 4096, 11008, 2), (700, 32859, 2048, 1), (3650, 22016, 1024, 3), (330, 4096, 11008, 2), (2500, 4096, 4096, 0)]


@pytest.mark.parametrize("m,n,k,epi", SK_SHAPES)
def test_gemm_stream_k_tail(lib, m, n, k, epi):
    """Target forwards of 4-64 users (260-4000 tokens): the ring kernel's tile grid is 0.2-3 rounds of 256 workgroups, and the k-steps of the
    last, partly filled round are dealt evenly over the chip (gemm_ring_kernel<..., SK>): parts of a tile meet through the workspace and the
    last one to arrive sums them in part order.  Checked on the Llama-7B projection shapes against torch fp32 on the same bf16 values, and:
    the sum must not depend on which part arrived last (three runs bit-identical), packed operands = row-major operands bit for bit."""
    a = _rand((m, k), 81, 1.0).to(torch.bfloat16).cuda()
    w = _rand((n, k), 82, 0.03).to(torch.bfloat16).cuda()
    ws = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    if epi == _lib.EPI_SWIGLU:
        from atspeed_amd.model import _interleave_gate_up
        w = _interleave_gate_up(w[: n // 2].contiguous(), w[n // 2:].contiguous())
    base = _rand((m, n), 83).to(torch.bfloat16).cuda() if epi == _lib.EPI_RESID else None
    if epi == _lib.EPI_F32:
        ldc = (n + 63) // 64 * 64; mk = lambda: torch.zeros(m, ldc, dtype=torch.float32, device="cuda")
    elif epi == _lib.EPI_SWIGLU:
        ldc = n // 2; mk = lambda: torch.zeros((m + 1) // 2 * 2, ldc, dtype=torch.bfloat16, device="cuda")
    else:
        ldc = n; mk = lambda: (base.clone() if epi == _lib.EPI_RESID else torch.zeros(m, n, dtype=torch.bfloat16, device="cuda"))
    outs = []
    # two parts per tail tile wherever they fit, whatever the cost model says; 257-384 tokens: the panel form, and 257-1100 tokens on N = 4096: the
    # K-cut form, would take some of these shapes first (process-wide switches: atspeed_set_switch, restored on exit)
    with _lib.switches(gemm_sk=2, gemm_panel=0, gemm_kcut=0):
        _path_counters(lib, reset=True)
        for _ in range(3):
            c = mk()
            _lib.check(lib.atspeed_gemm(a.data_ptr(), w.data_ptr(), c.data_ptr(), m, n, k, k, ldc, _lib.ATSPEED_BF16, epi, ws.data_ptr(), ws.numel(), _st()))
            outs.append(c)
        assert _path_counters(lib)[PATH_RING_SK] == 3, "this shape did not take the split-K tail: the test would pass on the plain ring kernel"
        ap, wp = _pack(lib, a), _pack(lib, w)
        cp = mk()
        _lib.check(lib.atspeed_gemm_packed(ap.data_ptr(), wp.data_ptr(), cp.data_ptr(), m, n, k, ldc, epi, ws.data_ptr(), ws.numel(), _st()))
        assert _path_counters(lib)[PATH_RING_SK] == 4
    cd = mk()                                                       # what the fitted cost model picks by itself (tail or not): same product, its own summation order
    _lib.check(lib.atspeed_gemm(a.data_ptr(), w.data_ptr(), cd.data_ptr(), m, n, k, k, ldc, _lib.ATSPEED_BF16, epi, ws.data_ptr(), ws.numel(), _st()))
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), "the stream-K sum depends on the arrival order"
    prod = a.float() @ w.float().T                                   # torch fp32 on the bf16 values
    if epi == _lib.EPI_SWIGLU:
        v = prod.view(m, n // 32, 2, 16)
        gate, up = v[:, :, 0].reshape(m, n // 2), v[:, :, 1].reshape(m, n // 2)
        ref = torch.nn.functional.silu(gate.to(torch.bfloat16).float()) * up.to(torch.bfloat16).float()
        got = outs[0][:m].float()
        assert torch.equal(_unpack(lib, cp, m), outs[0][:m])
        tol = 3e-2 * float(ref.abs().max())
    else:
        ref = prod + (base.float() if epi == _lib.EPI_RESID else 0.0)
        got = outs[0][:, :n].float()
        assert torch.equal(cp[:, :n], outs[0][:, :n])
        tol = (2e-5 * np.sqrt(k) if epi == _lib.EPI_F32 else 1e-2) * float(ref.abs().max())
    err = float((got - ref).abs().max())
    assert err <= tol, (err, tol)
    gd = cd[:m].float() if epi == _lib.EPI_SWIGLU else cd[:, :n].float()
    assert float((gd - ref).abs().max()) <= tol


@pytest.mark.parametrize("m,n,k,epi", [(912, 4096, 4096, 2), (912, 22016, 4096, 3), (1800, 12288, 4096, 0), (700, 32859, 2048, 1)])
def test_gemm_fp16_flavour_on_the_batched_paths(lib, m, n, k, epi):
    """The fp16 flavour (ATSPEED_F16: gemm.hip compiled for IEEE half, v_mfma_f32_16x16x32_f16) through the ring kernel and its split-K tail on
    the Llama-7B projection shapes: against torch fp32 on the same fp16 values, and tighter than bf16's bound (3 more significand bits)."""
    a = _rand((m, k), 91, 1.0).to(torch.float16).cuda()
    w = _rand((n, k), 92, 0.03).to(torch.float16).cuda()
    ws = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    if epi == _lib.EPI_SWIGLU:
        from atspeed_amd.model import _interleave_gate_up
        w = _interleave_gate_up(w[: n // 2].contiguous(), w[n // 2:].contiguous())
    base = _rand((m, n), 93).to(torch.float16).cuda() if epi == _lib.EPI_RESID else None
    if epi == _lib.EPI_F32:
        ldc = (n + 63) // 64 * 64; c = torch.zeros(m, ldc, dtype=torch.float32, device="cuda")
    elif epi == _lib.EPI_SWIGLU:
        ldc = n // 2; c = torch.zeros(m, ldc, dtype=torch.float16, device="cuda")
    else:
        ldc = n; c = base.clone() if epi == _lib.EPI_RESID else torch.zeros(m, n, dtype=torch.float16, device="cuda")
    _lib.check(lib.atspeed_gemm(a.data_ptr(), w.data_ptr(), c.data_ptr(), m, n, k, k, ldc, _lib.ATSPEED_F16, epi, ws.data_ptr(), ws.numel(), _st()))
    torch.cuda.synchronize()
    prod = a.float() @ w.float().T
    if epi == _lib.EPI_SWIGLU:
        v = prod.view(m, n // 32, 2, 16)
        gate, up = v[:, :, 0].reshape(m, n // 2), v[:, :, 1].reshape(m, n // 2)
        ref = torch.nn.functional.silu(gate.to(torch.float16).float()) * up.to(torch.float16).float()
        got, tol = c.float(), 4e-3 * float(ref.abs().max())
    else:
        ref = prod + (base.float() if epi == _lib.EPI_RESID else 0.0)
        got = c[:, :n].float()
        tol = (2e-5 * np.sqrt(k) if epi == _lib.EPI_F32 else 1.5e-3) * float(ref.abs().max())
    err = float((got - ref).abs().max())
    assert err <= tol, (err, tol)


def test_split_k_tail_from_two_streams_and_two_host_threads(lib):
    """VERDICT r4 #7 / ADVICE r4: callers without a model (this C ABI) share ONE split-K arena per device.  Launches on two streams, and from two
    host threads on their own streams, must give the single-stream bits: the library orders them (mutex around bookkeeping + enqueue, event
    from the previous stream).  912 x 4096 x 4096 and 912 x 22016 x 4096: thin / partly filled grids that take the tail."""
    import threading
    with _lib.switches(gemm_sk=2, gemm_kcut=0):
        jobs = []
        for i, (m, n, k, epi) in enumerate([(912, 4096, 4096, _lib.EPI_STORE), (912, 22016, 4096, _lib.EPI_STORE), (640, 4096, 11008, _lib.EPI_RESID)]):
            a = _rand((m, k), 181 + i, 1.0).to(torch.bfloat16).cuda()
            w = _rand((n, k), 191 + i, 0.03).to(torch.bfloat16).cuda()
            base = _rand((m, n), 171 + i).to(torch.bfloat16).cuda()
            jobs.append((a, w, base, m, n, k, epi))
        ws = [torch.empty(64 << 20, dtype=torch.uint8, device="cuda") for _ in range(2)]

        def run(job, stream, wsb):
            a, w, base, m, n, k, epi = job
            c = base.clone() if epi == _lib.EPI_RESID else torch.zeros(m, n, dtype=torch.bfloat16, device="cuda")
            stream.wait_stream(torch.cuda.current_stream())
            _lib.check(lib.atspeed_gemm(a.data_ptr(), w.data_ptr(), c.data_ptr(), m, n, k, k, n, _lib.ATSPEED_BF16, epi, wsb.data_ptr(), wsb.numel(), stream.cuda_stream))
            return c

        torch.cuda.synchronize()
        _path_counters(lib, reset=True)
        serial = [run(j, torch.cuda.current_stream(), ws[0]) for j in jobs]
        torch.cuda.synchronize()
        assert _path_counters(lib)[PATH_RING_SK] == len(jobs)
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        for rep in range(6):                                          # interleaved on two streams from one thread
            got = [run(j, (s1, s2)[(i + rep) & 1], ws[(i + rep) & 1]) for i, j in enumerate(jobs)]
            torch.cuda.synchronize()
            for g, want in zip(got, serial):
                assert torch.equal(g, want), "split-K tail: two streams disagree with one"
        results = [[], []]

        def worker(t):
            st = (s1, s2)[t]
            with torch.cuda.device(0):
                for rep in range(8):
                    results[t].append([run(j, st, ws[t]) for j in jobs])
            st.synchronize()

        th = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize()
        for t in range(2):
            for got in results[t]:
                for g, want in zip(got, serial):
                    assert torch.equal(g, want), "split-K tail: two host threads disagree with the serial result"


# ------------------------------------------------------------------ W8A8 in the reference's own regime: one user's forwards (1-256 rows)
FP8_SMALL = [(20, 12288, 4096, 0), (20, 4096, 4096, 2), (20, 22016, 4096, 3), (20, 4096, 11008, 2),            # the final single step (K beams)
             (60, 12288, 4096, 0), (100, 4096, 4096, 2), (121, 22016, 4096, 3), (121, 4096, 11008, 2),         # later rounds (K + dl * DK tokens)
             (228, 12288, 4096, 0), (228, 4096, 4096, 2), (228, 22016, 4096, 3), (228, 4096, 11008, 2),        # first verification (beamSD.py:221)
             (256, 22016, 4096, 3), (33, 2304, 768, 0), (77, 1000, 512, 1), (1, 4096, 4096, 2), (130, 3072, 1024, 3)]


@pytest.mark.parametrize("m,n,k,epi", FP8_SMALL)
def test_gemm_fp8_weight_streaming_form(lib, m, n, k, epi):
    """VERDICT r4 missing #1: the reference loads its target 8-bit for EVERY forward at batch 1 (inference.py:86-91), and m < 512 used to run
    bf16 here.  gemm_wdma_kernel<..., F8>: e4m3 rows through the weight-streaming ring, one v_mfma_scale_f32_16x16x128_f8f6f4 per tile and
    stage, per-row scales on the accumulators, narrow N cut in K (fp32 slabs + the 16-bit form's reduce).  Against the exact product of the
    dequantised operands (fp64), row-major and packed operands bit for bit, and the launch counters say which kernel ran."""
    x = _rand((m, k), 51, 1.5).to(torch.bfloat16).cuda()
    w = _rand((n, k), 52, 0.05).to(torch.bfloat16).cuda()
    xq = torch.empty(m, k, dtype=torch.uint8, device="cuda"); sx = torch.empty(m, device="cuda")
    wq = torch.empty(n, k, dtype=torch.uint8, device="cuda"); sw = torch.empty(n, device="cuda")
    _lib.check(lib.atspeed_quant_rows_fp8(x.data_ptr(), m, k, xq.data_ptr(), sx.data_ptr(), _st()))
    _lib.check(lib.atspeed_quant_rows_fp8(w.data_ptr(), n, k, wq.data_ptr(), sw.data_ptr(), _st()))
    ref = (_fp8_to_float(xq).double() @ _fp8_to_float(wq).double().T) * sx.cpu().double()[:, None] * sw.cpu().double()[None, :]
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    mp = (m + 1) // 2 * 2
    if epi == _lib.EPI_F32:
        ldc = (n + 63) // 64 * 64; mk = lambda: torch.zeros(mp, ldc, dtype=torch.float32, device="cuda")
    elif epi == _lib.EPI_SWIGLU:
        ldc = n // 2; mk = lambda: torch.zeros(mp, ldc, dtype=torch.bfloat16, device="cuda")
        g = ref.view(m, n // 32, 2, 16)
        ref = (torch.nn.functional.silu(g[:, :, 0]) * g[:, :, 1]).reshape(m, n // 2)
    else:
        ldc = n; base = torch.zeros(mp, n, dtype=torch.bfloat16, device="cuda")
        if epi == _lib.EPI_RESID:
            base[:m] = _rand((m, n), 53).to(torch.bfloat16).cuda()
            ref = ref + base[:m].double().cpu()
        mk = lambda: base.clone()
    _path_counters(lib, reset=True)
    c0, c1, c2 = mk(), mk(), mk()
    _lib.check(lib.atspeed_gemm_fp8(xq.data_ptr(), sx.data_ptr(), wq.data_ptr(), sw.data_ptr(), c0.data_ptr(), m, n, k, ldc, epi, ws.data_ptr(), ws.numel(), _st()))
    cnt = _path_counters(lib)
    tiles = (n + 63) // 64 if (n <= 16384 and n % 64 == 0) else (n + 127) // 128      # 64-row weight tiles up to N = 16384, else 128-row ones
    split = tiles < 150 and min(256 // tiles, (k // 128) // 4) >= 2                    # narrow projections are cut in K (at least 4 tiles of 128 k per part)
    assert cnt[PATH_FP8_WDMA_SPLIT if (split or epi == _lib.EPI_RESID) else PATH_FP8_WDMA] == 1 and cnt[PATH_FP8_RING] == 0, cnt
    xp, wp = _pack(lib, xq), _pack(lib, wq)
    _lib.check(lib.atspeed_gemm_fp8_packed(xp.data_ptr(), sx.data_ptr(), wp.data_ptr(), sw.data_ptr(), c1.data_ptr(), m, n, k, ldc, epi, ws.data_ptr(), ws.numel(), _st()))
    # without a workspace: one part per tile, same sums up to fp32 order; the residual epilogue, which has no one-part form in this kernel, takes the
    # ring kernel's (K % 256 == 0: every residual shape here; ADVICE r5 -- it returned ATSPEED_ERR_CAPACITY in round 5)
    _path_counters(lib, reset=True)
    _lib.check(lib.atspeed_gemm_fp8(xq.data_ptr(), sx.data_ptr(), wq.data_ptr(), sw.data_ptr(), c2.data_ptr(), m, n, k, ldc, epi, None, 0, _st()))
    if epi == _lib.EPI_RESID:
        assert k % 256 == 0 and _path_counters(lib)[PATH_FP8_RING] == 1
    torch.cuda.synchronize()
    out = c0.double().cpu()[:m, : ref.shape[1]]
    scale = float(ref.abs().max())
    tol = 2e-5 * scale * np.sqrt(k) if epi == _lib.EPI_F32 else 2e-2 * scale
    np.testing.assert_allclose(out.numpy(), ref.numpy(), atol=tol, rtol=0)
    if epi == _lib.EPI_SWIGLU:
        assert torch.equal(_unpack(lib, c1, m), c0[:m])
    else:
        assert torch.equal(c1[:m], c0[:m])
    np.testing.assert_allclose(c2.double().cpu()[:m, : ref.shape[1]].numpy(), ref.numpy(), atol=tol, rtol=0)


# ------------------------------------------------------------------ panel form of the ring kernel: 257-512 tokens in one launch
PANEL_SHAPES = [(320, 22016, 4096, 3), (320, 4096, 11008, 2), (320, 4096, 4096, 2), (384, 12288, 4096, 0), (257, 4096, 4096, 0), (384, 22016, 4096, 3),
                (383, 4096, 4096, 2), (300, 8192, 2048, 1), (320, 22016, 4096, 0), (352, 12288, 4096, 0)]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,n,k,epi", PANEL_SHAPES)
def test_gemm_panel_form(lib, m, n, k, epi, dtype):
    """VERDICT r4 missing #2 / weak #3: 257-512 tokens (4-16 users per lock-step batch, the K-token continuation forwards x users of
    beamSD.py:579-588; a long prompt's first verification, :221) used to take two 256-row token tiles of the ring kernel (38 % padding at 320
    tokens) or a thin grid with a split-K tail.  gemm_ring_kernel<..., WN = 2, WM = 4>: a workgroup owns a 128-row weight panel and all
    token rows (384 / 512); narrow projections are cut in K (fp32 slabs + the usual reduce).  Against torch fp32 on the same 16-bit values,
    packed = row-major bit for bit, two runs identical, and the launch counters say the panel kernel ran."""
    code = _lib.ATSPEED_BF16 if dtype == torch.bfloat16 else _lib.ATSPEED_F16
    a = _rand((m, k), 281, 1.0).to(dtype).cuda()
    w = _rand((n, k), 282, 0.03).to(dtype).cuda()
    ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    if epi == _lib.EPI_SWIGLU:
        from atspeed_amd.model import _interleave_gate_up
        w = _interleave_gate_up(w[: n // 2].contiguous(), w[n // 2:].contiguous())
    base = _rand((m, n), 283).to(dtype).cuda() if epi == _lib.EPI_RESID else None
    mp = (m + 1) // 2 * 2
    if epi == _lib.EPI_F32:
        ldc = (n + 63) // 64 * 64; mk = lambda: torch.zeros(m, ldc, dtype=torch.float32, device="cuda")
    elif epi == _lib.EPI_SWIGLU:
        ldc = n // 2; mk = lambda: torch.zeros(mp, ldc, dtype=dtype, device="cuda")
    else:
        ldc = n; mk = lambda: (base.clone() if epi == _lib.EPI_RESID else torch.zeros(m, n, dtype=dtype, device="cuda"))
    with _lib.switches(gemm_panel=2, gemm_kcut=0):                  # every shape the panel kernel can take, whatever the dispatch prefers
        _path_counters(lib, reset=True)
        outs = []
        for _ in range(2):
            c = mk()
            _lib.check(lib.atspeed_gemm(a.data_ptr(), w.data_ptr(), c.data_ptr(), m, n, k, k, ldc, code, epi, ws.data_ptr(), ws.numel(), _st()))
            outs.append(c)
        cnt = _path_counters(lib)
        split = (n + 127) // 128 < 150
        assert cnt[PATH_PANEL_SPLIT if split else PATH_PANEL] == 2 and cnt[PATH_RING] == cnt[PATH_RING_SK] == cnt[PATH_TILED] == 0, cnt
        torch.cuda.synchronize()
        assert torch.equal(outs[0], outs[1])
        if dtype == torch.bfloat16:                                 # the packed-operand entry point is the bf16 engine's
            ap, wp = _pack(lib, a), _pack(lib, w)
            cp = mk()
            _lib.check(lib.atspeed_gemm_packed(ap.data_ptr(), wp.data_ptr(), cp.data_ptr(), m, n, k, ldc, epi, ws.data_ptr(), ws.numel(), _st()))
            torch.cuda.synchronize()
    if dtype == torch.bfloat16:
        if epi == _lib.EPI_SWIGLU:
            assert torch.equal(_unpack(lib, cp, m), outs[0][:m])
        else:
            assert torch.equal(cp[:, :n], outs[0][:, :n])
    prod = a.float() @ w.float().T
    if epi == _lib.EPI_SWIGLU:
        v = prod.view(m, n // 32, 2, 16)
        gate, up = v[:, :, 0].reshape(m, n // 2), v[:, :, 1].reshape(m, n // 2)
        ref = torch.nn.functional.silu(gate.to(dtype).float()) * up.to(dtype).float()
        got = outs[0][:m].float()
        tol = (3e-2 if dtype == torch.bfloat16 else 5e-3) * float(ref.abs().max())
    else:
        ref = prod + (base.float() if epi == _lib.EPI_RESID else 0.0)
        got = outs[0][:, :n].float()
        tol = (2e-5 * np.sqrt(k) if epi == _lib.EPI_F32 else (1e-2 if dtype == torch.bfloat16 else 2e-3)) * float(ref.abs().max())
    err = float((got - ref).abs().max())
    assert err <= tol, (err, tol)


@pytest.mark.parametrize("m,n,k,epi", [(300, 4096, 4096, 2), (456, 4096, 11008, 2), (512, 4096, 4096, 2), (1000, 4096, 11008, 2), (912, 4096, 11008, 0), (640, 4096, 11008, 1),
                                      (300, 1024, 2048, 3)])
def test_gemm_fp8_ring_cut_in_k(lib, m, n, k, epi):
    """Round 5: the W8A8 projections whose 256-wide tile grid is thin (N = 4096 at 257-2000 tokens: o_proj and down of 2-8 users per lock-step batch)
    used to run on 32-80 workgroups (257-511 tokens) or fall back to bf16 (from 512).  gemm_ring_mx_kernel<..., SPLITK>: tiles x parts ~ one
    round, scaled fp32 slabs, the usual reduce.  Against the exact fp64 product of the dequantised operands; packed = row-major; counters."""
    x = _rand((m, k), 51, 1.5).to(torch.bfloat16).cuda()
    w = _rand((n, k), 52, 0.05).to(torch.bfloat16).cuda()
    xq = torch.empty(m, k, dtype=torch.uint8, device="cuda"); sx = torch.empty(m, device="cuda")
    wq = torch.empty(n, k, dtype=torch.uint8, device="cuda"); sw = torch.empty(n, device="cuda")
    _lib.check(lib.atspeed_quant_rows_fp8(x.data_ptr(), m, k, xq.data_ptr(), sx.data_ptr(), _st()))
    _lib.check(lib.atspeed_quant_rows_fp8(w.data_ptr(), n, k, wq.data_ptr(), sw.data_ptr(), _st()))
    ref = (_fp8_to_float(xq).double() @ _fp8_to_float(wq).double().T) * sx.cpu().double()[:, None] * sw.cpu().double()[None, :]
    ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    mp = (m + 1) // 2 * 2
    if epi == _lib.EPI_F32:
        ldc = (n + 63) // 64 * 64; mk = lambda: torch.zeros(mp, ldc, dtype=torch.float32, device="cuda")
    elif epi == _lib.EPI_SWIGLU:
        ldc = n // 2; mk = lambda: torch.zeros(mp, ldc, dtype=torch.bfloat16, device="cuda")
        g = ref.view(m, n // 32, 2, 16)
        ref = (torch.nn.functional.silu(g[:, :, 0]) * g[:, :, 1]).reshape(m, n // 2)
    else:
        ldc = n; base = torch.zeros(mp, n, dtype=torch.bfloat16, device="cuda")
        if epi == _lib.EPI_RESID:
            base[:m] = _rand((m, n), 53).to(torch.bfloat16).cuda()
            ref = ref + base[:m].double().cpu()
        mk = lambda: base.clone()
    _path_counters(lib, reset=True)
    c0, c1, c2 = mk(), mk(), mk()
    _lib.check(lib.atspeed_gemm_fp8(xq.data_ptr(), sx.data_ptr(), wq.data_ptr(), sw.data_ptr(), c0.data_ptr(), m, n, k, ldc, epi, ws.data_ptr(), ws.numel(), _st()))
    cnt = _path_counters(lib)
    assert cnt[11] == 1 and cnt[PATH_FP8_RING] == 0, cnt                  # 11: fp8 ring kernel cut in K
    xp, wp = _pack(lib, xq), _pack(lib, wq)
    _lib.check(lib.atspeed_gemm_fp8_packed(xp.data_ptr(), sx.data_ptr(), wp.data_ptr(), sw.data_ptr(), c1.data_ptr(), m, n, k, ldc, epi, ws.data_ptr(), ws.numel(), _st()))
    _lib.check(lib.atspeed_gemm_fp8(xq.data_ptr(), sx.data_ptr(), wq.data_ptr(), sw.data_ptr(), c2.data_ptr(), m, n, k, ldc, epi, None, 0, _st()))   # no workspace: the plain grid
    assert _path_counters(lib)[PATH_FP8_RING] == 1
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    tol = 2e-5 * scale * np.sqrt(k) if epi == _lib.EPI_F32 else 2e-2 * scale
    for c in (c0, c2):
        np.testing.assert_allclose(c.double().cpu()[:m, : ref.shape[1]].numpy(), ref.numpy(), atol=tol, rtol=0)
    if epi == _lib.EPI_SWIGLU:
        assert torch.equal(_unpack(lib, c1, m), c0[:m])
    else:
        assert torch.equal(c1[:m], c0[:m])


@pytest.mark.parametrize("m,n,k,epi,G", [(900, 4096, 4096, 2, 150), (912, 22016, 4096, 3, 201), (400, 4096, 4096, 0, 100), (640, 4096, 11008, 2, 90), (700, 32859, 2048, 1, 10)])
def test_split_k_tail_with_an_unaligned_deal(lib, m, n, k, epi, G):
    """the ring kernel's tail accepts ANY even deal of the tail's k-units over G workgroups (a workgroup may end one tile and start the next:
    the segment loop, the second slot), but the launcher only makes aligned plans, so that code ran in no test.  the test hook `gemm_sk_g` (atspeed_set_switch)
    forces an unaligned G: same product as torch fp32 on the bf16 values, bit-identical across runs (the sum must not depend on the arrival order)."""
    a = _rand((m, k), 381, 1.0).to(torch.bfloat16).cuda()
    w = _rand((n, k), 382, 0.03).to(torch.bfloat16).cuda()
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    if epi == _lib.EPI_SWIGLU:
        from atspeed_amd.model import _interleave_gate_up
        w = _interleave_gate_up(w[: n // 2].contiguous(), w[n // 2:].contiguous())
    base = _rand((m, n), 383).to(torch.bfloat16).cuda() if epi == _lib.EPI_RESID else None
    if epi == _lib.EPI_F32:
        ldc = (n + 63) // 64 * 64; mk = lambda: torch.zeros(m, ldc, dtype=torch.float32, device="cuda")
    elif epi == _lib.EPI_SWIGLU:
        ldc = n // 2; mk = lambda: torch.zeros(m, ldc, dtype=torch.bfloat16, device="cuda")
    else:
        ldc = n; mk = lambda: (base.clone() if epi == _lib.EPI_RESID else torch.zeros(m, n, dtype=torch.bfloat16, device="cuda"))
    with _lib.switches(gemm_sk=2, gemm_panel=0, gemm_kcut=0, gemm_sk_g=G):
        _path_counters(lib, reset=True)
        outs = []
        for _ in range(3):
            c = mk()
            _lib.check(lib.atspeed_gemm(a.data_ptr(), w.data_ptr(), c.data_ptr(), m, n, k, k, ldc, _lib.ATSPEED_BF16, epi, ws.data_ptr(), ws.numel(), _st()))
            outs.append(c)
        assert _path_counters(lib)[PATH_RING_SK] == 3
        torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    prod = a.float() @ w.float().T
    if epi == _lib.EPI_SWIGLU:
        v = prod.view(m, n // 32, 2, 16)
        ref = torch.nn.functional.silu(v[:, :, 0].reshape(m, n // 2).to(torch.bfloat16).float()) * v[:, :, 1].reshape(m, n // 2).to(torch.bfloat16).float()
        got, tol = outs[0].float(), 3e-2 * float(ref.abs().max())
    else:
        ref = prod + (base.float() if epi == _lib.EPI_RESID else 0.0)
        got, tol = outs[0][:, :n].float(), (2e-5 * np.sqrt(k) if epi == _lib.EPI_F32 else 1e-2) * float(ref.abs().max())
    assert float((got - ref).abs().max()) <= tol

# ------------------------------------------------------------------ round 6: K-cut ring form
KCUT_SHAPES = [(320, 4096, 4096, 2), (320, 4096, 11008, 2), (640, 4096, 4096, 2), (640, 4096, 11008, 2), (1100, 4096, 11008, 2), (1054, 4096, 4096, 2),
               (257, 4096, 4096, 0), (900, 4096, 4096, 1), (700, 2048, 8192, 0), (600, 3000, 2048, 2), (1099, 1028, 2304, 0)]


@pytest.mark.parametrize("m,n,k,epi,dtype", [s + (torch.bfloat16,) for s in KCUT_SHAPES] + [s + (torch.float16,) for s in KCUT_SHAPES[1:6:2]])
def test_gemm_ring_cut_in_k(lib, m, n, k, epi, dtype):
    """Round 6: the 16-bit N <= 4096 projections at 257-1100 tokens (o_proj, down of 4-16 users in lock step) run the ring kernel cut in K over the
    whole chip -- tiles x parts ~ 256 workgroups, fp32 slabs, the reduce kernels of the other split forms -- as the default dispatch
    (`gemm_kcut`, 2: also where the panel form applied).  Against torch fp32 on the same values, packed = row-major bit for bit, two runs
    identical, and the counters say the split ring kernel ran and neither the tail, the panel nor the LDS-tiled kernel."""
    code = _lib.ATSPEED_BF16 if dtype == torch.bfloat16 else _lib.ATSPEED_F16
    a = _rand((m, k), 481, 1.0).to(dtype).cuda()
    w = _rand((n, k), 482, 0.03).to(dtype).cuda()
    ws = torch.empty(128 << 20, dtype=torch.uint8, device="cuda")
    base = _rand((m, n), 483).to(dtype).cuda() if epi == _lib.EPI_RESID else None
    if epi == _lib.EPI_F32:
        ldc = (n + 63) // 64 * 64; mk = lambda: torch.zeros(m, ldc, dtype=torch.float32, device="cuda")
    else:
        ldc = n; mk = lambda: (base.clone() if epi == _lib.EPI_RESID else torch.zeros(m, n, dtype=dtype, device="cuda"))
    _path_counters(lib, reset=True)
    outs = []
    for _ in range(2):
        c = mk()
        _lib.check(lib.atspeed_gemm(a.data_ptr(), w.data_ptr(), c.data_ptr(), m, n, k, k, ldc, code, epi, ws.data_ptr(), ws.numel(), _st()))
        outs.append(c)
    cnt = _path_counters(lib)
    assert cnt[PATH_RING_SPLIT] == 2 and cnt[PATH_RING] == cnt[PATH_RING_SK] == cnt[PATH_PANEL] == cnt[PATH_PANEL_SPLIT] == cnt[PATH_TILED] == 0, cnt
    if dtype == torch.bfloat16:
        ap, wp = _pack(lib, a), _pack(lib, w)
        cp = mk()
        _lib.check(lib.atspeed_gemm_packed(ap.data_ptr(), wp.data_ptr(), cp.data_ptr(), m, n, k, ldc, epi, ws.data_ptr(), ws.numel(), _st()))
    with _lib.switches(gemm_kcut=0):                                 # what ran before: same product, its own summation order
        c0 = mk()
        _lib.check(lib.atspeed_gemm(a.data_ptr(), w.data_ptr(), c0.data_ptr(), m, n, k, k, ldc, code, epi, ws.data_ptr(), ws.numel(), _st()))
    torch.cuda.synchronize()
    assert _path_counters(lib)[PATH_RING_SPLIT] == (3 if dtype == torch.bfloat16 else 2)
    assert torch.equal(outs[0], outs[1])
    if dtype == torch.bfloat16:
        assert torch.equal(cp, outs[0])
    ref = a.float() @ w.float().T + (base.float() if epi == _lib.EPI_RESID else 0.0)
    got = outs[0][:, :n].float()
    tol = (2e-5 * np.sqrt(k) if epi == _lib.EPI_F32 else 1e-2) * float(ref.abs().max())
    assert float((got - ref).abs().max()) <= tol and float((c0[:, :n].float() - ref).abs().max()) <= tol
