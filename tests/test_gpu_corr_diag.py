"""The diagonal-sliding candidate kernel of the correlation (csrc/corr_diag16.hip) against the float64 oracle of
model/SearchTransfer.py:26-34 (normalised 3x3 unfold, bmm, max over the reference positions) and against the slab kernel it
replaces: ragged maps (partial tiles, heights that are not a multiple of the 4 diagonals of a workgroup, maps smaller than one
tile, single rows), reference maps of another size than the query map (higher: the diagonals are cyclic in the reference height;
SelfTransfer's rotated map), exact ties, walks cut into segments."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from speinet_amd.ops import Ctx, FMap  # noqa: E402

DEV = "cuda:0"


def rnd(seed, *shape):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def fm(x):
    n, c, h, w = x.shape
    assert n == 1
    return FMap(x[0].permute(1, 2, 0).reshape(h * w, c).contiguous().to(DEV), h, w, c)


def oracle_top2(lr3, rf3):
    lu = F.normalize(F.unfold(lr3, (3, 3), padding=1), dim=1)
    ru = F.normalize(F.unfold(rf3, (3, 3), padding=1).permute(0, 2, 1), dim=2)
    r = torch.bmm(ru.double(), lu.double())[0]                       # [Nr, Nl]
    return r, torch.topk(r, min(2, r.shape[0]), dim=0)


@pytest.mark.parametrize("h,w", [(37, 45), (20, 30), (64, 64), (9, 130), (1, 5), (3, 3), (2, 70), (50, 50), (45, 80), (4, 64), (5, 65)])
@pytest.mark.parametrize("mode", ["bf16", "f16"])
def test_corr_diag_vs_oracle(h, w, mode):
    lr3, rf3 = rnd(100 + h, 1, 128, h, w), rnd(200 + w, 1, 128, h, w)
    r, top = oracle_top2(lr3, rf3)
    ops = Ctx(mode, "top2", device=DEV)
    inv_l, inv_r = ops.patch_invnorm(fm(lr3)), ops.patch_invnorm(fm(rf3))
    plan = ops.corr_plan(fm(lr3), fm(rf3), inv_l, inv_r)
    assert plan.kernel.startswith("corr_diag_kernel")
    plan.launch()
    s, arg = plan.s, plan.arg
    diff = arg.cpu().long() != top.indices[0]
    if top.values.shape[0] > 1:
        margin = (top.values[0] - top.values[1])[diff]
        assert diff.sum().item() <= 2 and (margin < 1e-6).all(), (mode, diff.sum().item(), margin)
    else:
        assert not diff.any()
    assert (s.cpu().double() - top.values[0]).abs().max().item() < 1e-6
    # same decision as the slab kernel's candidate pass + re-score
    ops0 = ops.replace(corr_diag=False)
    plan0 = ops0.corr_plan(fm(lr3), fm(rf3), inv_l, inv_r)
    assert plan0.kernel.startswith("corr_slab_kernel")
    plan0.launch()
    assert (plan0.arg != arg).sum().item() <= 2


@pytest.mark.parametrize("mode", ["bf16", "f16"])
def test_corr_diag_exact_ties(mode):
    """Many bit-identical reference patches (a constant reference map away from its border, and a query map tiled from one
    row): the lowest index among the exact maxima must win, as torch.max does.  Positive features, so that every score is
    positive: the position tag in the low mantissa bits of a candidate key orders equal scores by position only then (a
    negative maximum with exact ties is broken towards the HIGHER position by both candidate kernels; the re-score still
    returns a maximal position)."""
    h, w = 22, 70
    lr3 = rnd(7, 1, 128, 1, w).abs().expand(1, 128, h, w).contiguous()
    rf3 = rnd(8, 1, 128, 1, 1).abs().expand(1, 128, h, w).contiguous()
    r, top = oracle_top2(lr3, rf3)
    ops = Ctx(mode, "top2", device=DEV)
    inv_l, inv_r = ops.patch_invnorm(fm(lr3)), ops.patch_invnorm(fm(rf3))
    s, arg = ops.corr_argmax(fm(lr3), fm(rf3), inv_l, inv_r)
    s0, arg0 = ops.replace(corr_diag=False).corr_argmax(fm(lr3), fm(rf3), inv_l, inv_r)
    # the slab kernel compares candidate keys WITH their in-block position tag across reference blocks, so among exact ties it may
    # keep a later block's first position; this kernel compares (score, index): same maximum, never a higher index
    assert (s - s0).abs().max().item() < 1e-6 and (arg <= arg0).all()
    # the lowest index among the positions whose float64 score equals the maximum (to round-off)
    first = (r >= r.max(dim=0, keepdim=True).values - 1e-12).float().argmax(dim=0)
    assert torch.equal(arg.cpu().long(), first)


def test_corr_diag_candidates_contain_argmax_720p_rows():
    """A 720p-wide map (5 tiles per row, walk cut into segments): the re-scored winner equals the slab kernel's everywhere."""
    h, w = 48, 320
    lr3, rf3 = rnd(31, 1, 128, h, w), rnd(32, 1, 128, h, w)
    ops = Ctx("f16", "top2", device=DEV)
    inv_l, inv_r = ops.patch_invnorm(fm(lr3)), ops.patch_invnorm(fm(rf3))
    s, arg = ops.corr_argmax(fm(lr3), fm(rf3), inv_l, inv_r)
    s0, arg0 = ops.replace(corr_diag=False).corr_argmax(fm(lr3), fm(rf3), inv_l, inv_r)
    assert (arg != arg0).sum().item() <= 2
    assert (s - s0).abs().max().item() < 1e-6


@pytest.mark.parametrize("hl,wl,hr,wr", [(29, 51, 37, 45), (20, 70, 70, 20), (45, 80, 80, 45), (1, 5, 9, 3), (12, 130, 13, 66), (37, 45, 37, 70)])
@pytest.mark.parametrize("mode", ["bf16", "f16"])
def test_corr_diag_other_reference_size(hl, wl, hr, wr, mode):
    """Reference map of another size, at least as high as the query map (hr >= hl) — incl. the transposed shapes SelfTransfer
    produces for landscape frames (model/SearchTransfer.py:60)."""
    lr3, rf3 = rnd(300 + hl, 1, 128, hl, wl), rnd(400 + wr, 1, 128, hr, wr)
    r, top = oracle_top2(lr3, rf3)
    ops = Ctx(mode, "top2", device=DEV)
    inv_l, inv_r = ops.patch_invnorm(fm(lr3)), ops.patch_invnorm(fm(rf3))
    plan = ops.corr_plan(fm(lr3), fm(rf3), inv_l, inv_r)
    assert plan.kernel.startswith("corr_diag_kernel")
    plan.launch()
    diff = plan.arg.cpu().long() != top.indices[0]
    margin = (top.values[0] - top.values[1])[diff]
    assert diff.sum().item() <= 2 and (margin < 1e-6).all(), (mode, diff.sum().item(), margin)
    assert (plan.s.cpu().double() - top.values[0]).abs().max().item() < 1e-6


def test_corr_diag_lower_reference_map_takes_the_slab_kernel():
    """hr < hl: the cyclic walk would leave the reference map; ops falls back to the slab kernel and the C entry refuses."""
    from speinet_amd import _lib
    lr3, rf3 = rnd(61, 1, 128, 37, 45), rnd(62, 1, 128, 29, 51)
    ops = Ctx("f16", "top2", device=DEV)
    inv_l, inv_r = ops.patch_invnorm(fm(lr3)), ops.patch_invnorm(fm(rf3))
    plan = ops.corr_plan(fm(lr3), fm(rf3), inv_l, inv_r)
    assert plan.kernel.startswith("corr_slab_kernel")
    n = 37 * 45
    z = torch.zeros(n, device=DEV)
    zi = torch.zeros(n, device=DEV, dtype=torch.int32)
    h16 = torch.zeros(n, 128, device=DEV, dtype=torch.float16)
    ws = torch.zeros(int(_lib.lib().spei_corr_diag_ws_floats(37, 45, 29, 51)), device=DEV)
    tp = ops._tp
    rc = _lib.lib().spei_corr_diag_top2_16(2, tp(h16), tp(h16), tp(inv_r), 37, 45, 29, 51, 128, tp(z), tp(zi), tp(z), tp(zi), tp(ws), ops._stream())
    assert rc != 0 and b"Hr" in _lib.lib().spei_last_error()


def test_corr_diag_workspace_budget_falls_back_to_the_slab_kernel(monkeypatch):
    """The diagonal kernel's candidate workspace grows with Hr * Wr / 64 * Hl * Wl (34 GB at 4K): beyond `ops.CORR_DIAG_WS_MAX` the
    slab kernel (workspace per query only) takes the call, with the same decision."""
    from speinet_amd import ops as ops_mod
    lr3, rf3 = rnd(71, 1, 128, 24, 40), rnd(72, 1, 128, 24, 40)
    ops = Ctx("f16", "top2", device=DEV)
    inv_l, inv_r = ops.patch_invnorm(fm(lr3)), ops.patch_invnorm(fm(rf3))
    s, arg = ops.corr_argmax(fm(lr3), fm(rf3), inv_l, inv_r)
    monkeypatch.setattr(ops_mod, "CORR_DIAG_WS_MAX", 1 << 10)
    plan = ops.corr_plan(fm(lr3), fm(rf3), inv_l, inv_r)
    assert plan.kernel.startswith("corr_slab_kernel")
    plan.launch()
    assert torch.equal(plan.arg, arg) and (plan.s - s).abs().max().item() < 1e-6


def test_corr_diag_random_shapes():
    """A seeded sweep of 24 ragged shape pairs (query hl x wl, reference hr x wr with hr >= hl): segment boundaries, wraps of the cyclic
    diagonals inside a segment, partial tiles on either map, groups of diagonals that are not full — against the float64 oracle."""
    import random
    rng = random.Random(20260403)
    ops = Ctx("f16", "top2", device=DEV)
    for case in range(24):
        hl, wl = rng.randint(1, 60), rng.randint(1, 140)
        hr, wr = hl + rng.choice((0, 0, 1, 3, 17, 40)), rng.choice((wl, rng.randint(1, 140)))
        lr3, rf3 = rnd(1000 + case, 1, 128, hl, wl), rnd(2000 + case, 1, 128, hr, wr)
        r, top = oracle_top2(lr3, rf3)
        inv_l, inv_r = ops.patch_invnorm(fm(lr3)), ops.patch_invnorm(fm(rf3))
        plan = ops.corr_plan(fm(lr3), fm(rf3), inv_l, inv_r)
        assert plan.kernel.startswith("corr_diag_kernel")
        plan.launch()
        arg = plan.arg.cpu().long()
        chosen = r.gather(0, arg.view(1, -1))[0]                      # float64 score of the position the kernels chose
        # S is the exact score of the chosen position (fp64 re-score); the choice is the maximum except where three or more positions
        # lie within the noise of the 16-bit candidate scores (~5e-5 for white-noise features: the true winner was not among the two
        # candidates) — then the chosen score is within that noise of the maximum
        assert (plan.s.cpu().double() - chosen).abs().max().item() < 1e-6, (case, (hl, wl, hr, wr))
        gap = top.values[0] - chosen
        nflip = int((arg != top.indices[0]).sum())
        assert gap.max().item() < 2e-4 and nflip <= max(2, arg.numel() // 1000), (case, (hl, wl, hr, wr), nflip, gap.max().item())
