"""BASELINE configs 2 / 3 at their full size on the GPU: 4096 hypotheses over 64 synthetic 1024x512 panoramas through the
fused pipeline, in TWO launch shapes:

* `test_benchmark_launch_shape_against_the_oracle`: exactly bench.py's default -- RenderVerifyPipeline(chunk=4096, one HIP
  stream), ONE launch of 4096 renders / samples per stage (the densify grid, the verifier batch and the auto-selection of the
  8-phase convolution kernel that only this shape has).  32 randomly chosen hypotheses are checked against the oracle end to
  end: BEV image bit for bit, and the LOGITS of the batch-4096 forward against oracle/resnet_oracle.forward on the oracle's
  own fp32 tiles, absolute 1e-3 (north_star's bound).
* `test_config2_config3_at_benchmark_size`: chunks of 1024 on three HIP streams (the overlapped schedule) -- 32 hypotheses
  bit for bit against the oracle (pixel index of every panorama point, final BEV image) from the overlapped run's own
  buffers, and all 4096 logits and the resident BEV images identical between the three-stream and the one-stream schedule:
  the work distribution inside the densify kernel and the triangle cache are timing dependent, the results must not be.
The oracle renders (about 1.5 s each) run in a spawned worker pool that never touches the GPU.
"""

import multiprocessing as mp
from types import SimpleNamespace

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from salve_amd import synthetic  # noqa: E402
from salve_amd.rasteriser import pack_hypotheses  # noqa: E402

N_HYP, N_PANOS, CHUNK = 4096, 64, 1024


def test_config2_config3_at_benchmark_size():
    from salve_amd.models.early_fusion import EarlyFusionCEResnet
    from salve_amd.pipeline import RenderVerifyPipeline

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = EarlyFusionCEResnet(50, False, 2, SimpleNamespace(modalities=["floor_rgb_texture"])).eval()
    from _helpers import oracle_floor_render, randomise_bn

    randomise_bn(model)
    panos = [synthetic.make_pano(i) for i in range(N_PANOS)]
    rgb, depth = np.stack([p[0] for p in panos]), np.stack([p[1] for p in panos])
    hyp = synthetic.make_hypotheses(N_HYP, N_PANOS, seed=0)
    rng = np.random.default_rng(5)
    # the BEV images still in the pipeline's buffers after a pass are those of the last two chunks
    picked = np.concatenate([rng.choice(np.arange(2 * CHUNK, 3 * CHUNK), 16, replace=False), rng.choice(np.arange(3 * CHUNK, 4 * CHUNK), 16, replace=False)])
    with mp.get_context("spawn").Pool(8) as pool:
        pending = pool.map_async(oracle_floor_render, [(hyp.i1[j], hyp.R[j], hyp.t[j]) for j in picked])

        results = {}
        for streams, overlap in ((3, True), (1, False)):
            pipe = RenderVerifyPipeline(model, dev, chunk=CHUNK, overlap=overlap, streams=streams)
            pipe.load_panos(rgb, depth)
            prepared = pipe.prepare(hyp)
            logits = pipe.score(prepared)
            torch.cuda.synchronize()
            pipe.check(f"full size, {streams} stream(s)")
            assert pipe.valid_mask(prepared).all()
            # the BEV images still resident after the pass: those of the last chunk (one buffer set) or the last two
            last = sorted(pipe.last_chunk_buffer)[-pipe.nbuf:]
            bev_last = {ci: pipe.bevs[pipe.last_chunk_buffer[ci]].clone() for ci in last}
            results[streams] = (logits.clone(), bev_last, pipe, prepared)
        oracle = pending.get(timeout=900)

    l3, b3, pipe3, prep3 = results[3]
    l1, b1, _, _ = results[1]
    assert torch.isfinite(l3).all()
    assert torch.equal(l3, l1), "logits differ between the three-stream and the one-stream schedule"
    assert torch.equal(b3[3], b1[3]), "BEV images of the last chunk differ between the schedules"

    # bit-exact against the oracle, from the overlapped run's own buffers
    ras = pipe3.ras
    for j, (bev_exp, xy_exp) in zip(picked, oracle):
        chunk, slot = pipe3.bev_index(prep3, int(j))   # (renders are issued in panorama order inside a chunk: pipeline.prepare)
        assert chunk == int(j) // CHUNK
        got = ras.export_u8(b3[chunk][slot:slot + 1]).cpu().numpy()[0]
        assert bev_exp is not None
        assert np.array_equal(got, bev_exp), f"hypothesis {j}: BEV image differs from the oracle"
        # indices: re-render this hypothesis alone with the debug outputs
        row = ras.upload_hypotheses(pack_hypotheses([hyp.i1[j]], [0], hyp.R[j:j + 1], hyp.t[j:j + 1], [1]))
        _, dbg = ras.render(pipe3.pano_rgb, pipe3.pano_depth, row, 1, debug=True)
        xy = dbg.img_xy[0].cpu().numpy()
        assert np.array_equal(xy[xy[:, 0] >= 0], xy_exp), f"hypothesis {j}: pixel indices differ"
        assert int(dbg.in_window[0]) == xy_exp.shape[0]


def _oracle_pair_logits(args):
    """(pool worker, CPU only) both oracle renders of one hypothesis -> (BEV image of the posed render, fp32 logits of the
    oracle's ResNet-50 on the oracle's fp32 tiles).  The weights are rebuilt from the same seeds as the test's model."""
    i1, i2, R, t = args
    import torch as th

    th.set_num_threads(1)
    from oracle import bev_oracle as bo
    from oracle import resnet_oracle as ro
    from salve_amd import synthetic as syn
    from salve_amd.models.early_fusion import EarlyFusionCEResnet
    from _helpers import randomise_bn

    th.manual_seed(0)
    model = EarlyFusionCEResnet(50, False, 2, SimpleNamespace(modalities=["floor_rgb_texture"])).eval()
    randomise_bn(model)
    p1, p2 = syn.make_pano(int(i1)), syn.make_pano(int(i2))
    r1, r2 = bo.render_bev_pair(p1[0], p1[1], p2[0], p2[1], R, t, "floor", mode="exact")
    x = th.from_numpy(np.concatenate([bo.tile_from_bev(r1["bev"]), bo.tile_from_bev(r2["bev"])], 0))[None]
    with th.no_grad():
        ref = ro.forward(model.state_dict(), 50, [x[:, :3], x[:, 3:]])
    return r1["bev"], ref[0].numpy()


def test_benchmark_launch_shape_against_the_oracle():
    from salve_amd.models.early_fusion import EarlyFusionCEResnet
    from salve_amd.pipeline import RenderVerifyPipeline

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = EarlyFusionCEResnet(50, False, 2, SimpleNamespace(modalities=["floor_rgb_texture"])).eval()
    from _helpers import randomise_bn

    randomise_bn(model)
    panos = [synthetic.make_pano(i) for i in range(N_PANOS)]
    rgb, depth = np.stack([p[0] for p in panos]), np.stack([p[1] for p in panos])
    hyp = synthetic.make_hypotheses(N_HYP, N_PANOS, seed=0)
    picked = np.random.default_rng(11).choice(N_HYP, 32, replace=False)
    with mp.get_context("spawn").Pool(8) as pool:
        pending = pool.map_async(_oracle_pair_logits, [(hyp.i1[j], hyp.i2[j], hyp.R[j], hyp.t[j]) for j in picked])
        pipe = RenderVerifyPipeline(model, dev, chunk=N_HYP, overlap=False, streams=1)   # bench.py's defaults
        pipe.load_panos(rgb, depth)
        prepared = pipe.prepare(hyp)
        logits = pipe.score(prepared)
        torch.cuda.synchronize()
        pipe.check("benchmark launch shape")
        assert pipe.valid_mask(prepared).all() and torch.isfinite(logits).all()
        where = np.array([pipe.bev_index(prepared, int(j))[1] for j in np.sort(picked)])   # render order != hypothesis order (pipeline.prepare)
        got_bev = pipe.ras.export_u8(pipe.bevs[0][torch.from_numpy(where).to(dev)]).cpu().numpy()
        got_logits = logits.cpu().numpy()
        oracle = pending.get(timeout=1200)
    order = {int(j): k for k, j in enumerate(np.sort(picked))}
    worst = 0.0
    for j, (bev_exp, logit_exp) in zip(picked, oracle):
        assert np.array_equal(got_bev[order[int(j)]], bev_exp), f"hypothesis {j}: BEV image of the 4096-render launch differs from the oracle"
        err = float(np.abs(got_logits[j] - logit_exp).max())
        worst = max(worst, err)
        assert err <= 1e-3, f"hypothesis {j}: logits of the batch-4096 forward {got_logits[j]} vs oracle {logit_exp}"
    print(f"benchmark launch shape (4096 per launch, one stream): 32 hypotheses, BEV bit-exact, max |logit - oracle| {worst:.2e}")


def test_config4_table_as_eight_serial_shards():
    """BASELINE config 4's WORKLOAD on one GPU: the 32 768-row table in the 8 contiguous blocks of 4096 the 8 ranks would own
    (reference work list scripts/render_dataset_bev.py:91-117; gather of the model outputs train_utils.py:214-215), every block
    scored through RenderVerifyPipeline(chunk=4096) and collected with `gather_logits(force=True)` -- the RCCL
    `all_gather_into_tensor` in a world of one.  The concatenation must equal one pass over the unsharded table bit for bit, and
    16 random rows are checked against the oracle: BEV image bit for bit, logits absolute 1e-3."""
    import torch.distributed as dist

    from salve_amd.models.early_fusion import EarlyFusionCEResnet
    from salve_amd.pipeline import RenderVerifyPipeline, gather_logits
    from _helpers import randomise_bn

    N, G = 32768, 8
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = EarlyFusionCEResnet(50, False, 2, SimpleNamespace(modalities=["floor_rgb_texture"])).eval()
    randomise_bn(model)
    panos = [synthetic.make_pano(i) for i in range(N_PANOS)]
    rgb, depth = np.stack([p[0] for p in panos]), np.stack([p[1] for p in panos])
    table = synthetic.make_hypotheses(N, N_PANOS, seed=0)
    picked = np.random.default_rng(4).choice(N, 16, replace=False)
    own_group = not dist.is_initialized()
    if own_group:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29541", rank=0, world_size=1, device_id=dev)
    try:
        with mp.get_context("spawn").Pool(8) as pool:
            pending = pool.map_async(_oracle_pair_logits, [(table.i1[j], table.i2[j], table.R[j], table.t[j]) for j in picked])
            pipe = RenderVerifyPipeline(model, dev, chunk=4096, overlap=False, streams=1)
            pipe.load_panos(rgb, depth)
            parts, bevs = [], {}
            for r in range(G):
                lo, hi = table.shard_bounds(N, r, G)
                assert hi - lo == 4096
                prepared = pipe.prepare(table.shard(r, G))
                local = pipe.score(prepared)
                torch.cuda.synchronize()
                pipe.check(f"config 4, shard {r}")
                assert pipe.valid_mask(prepared).all()
                parts.append(gather_logits(local, 1, force=True).clone())     # RCCL, world of one
                for j in picked[(picked >= lo) & (picked < hi)]:
                    k = pipe.bev_index(prepared, int(j - lo))[1]
                    bevs[int(j)] = pipe.ras.export_u8(pipe.bevs[0][k:k + 1]).cpu().numpy()[0]
            sharded = torch.cat(parts, 0)
            # one pass over the unsharded table (eight chunks of 4096 inside ONE score() call)
            prepared = pipe.prepare(table)
            whole = pipe.score(prepared)
            torch.cuda.synchronize()
            pipe.check("config 4, unsharded")
            assert torch.isfinite(whole).all()
            assert torch.equal(sharded, whole), "logits of the eight shards differ from one pass over the whole table"
            got = sharded.cpu().numpy()
            oracle = pending.get(timeout=1200)
    finally:
        if own_group:
            dist.destroy_process_group()
    worst = 0.0
    for j, (bev_exp, logit_exp) in zip(picked, oracle):
        assert np.array_equal(bevs[int(j)], bev_exp), f"row {j}: BEV image differs from the oracle"
        err = float(np.abs(got[j] - logit_exp).max())
        worst = max(worst, err)
        assert err <= 1e-3, f"row {j}: logits {got[j]} vs oracle {logit_exp}"
    print(f"config 4 workload (32768 rows as 8 serial shards of 4096 + world-1 RCCL gather): == unsharded pass bit for bit; "
          f"16 rows vs oracle: BEV bit-exact, max |logit - oracle| {worst:.2e}")


def _oracle_config5_hypothesis(args):
    """(pool worker, CPU only) BASELINE config 5, one hypothesis: the four oracle renders (ceiling and floor of both 2048 x 1024
    panoramas) -> (BEV images of the two posed renders, fp32 logits of the oracle's 12-channel ResNet-152 on the oracle's fp32 tiles)."""
    i1, i2, R, t = args
    import torch as th

    th.set_num_threads(2)
    from oracle import bev_oracle as bo
    from oracle import resnet_oracle as ro
    from salve_amd import synthetic as syn
    from salve_amd.models.early_fusion import EarlyFusionCEResnet
    from _helpers import randomise_bn

    th.manual_seed(0)
    model = EarlyFusionCEResnet(152, False, 2, SimpleNamespace(modalities=["ceiling_rgb_texture", "floor_rgb_texture"])).eval()
    randomise_bn(model)
    p1, p2 = syn.make_pano(int(i1), 1024, 2048), syn.make_pano(int(i2), 1024, 2048)
    bevs, tiles = [], []
    for surface in ("ceiling", "floor"):   # channel order of the 4-tuple: zind_data.py:306-315
        r1, r2 = bo.render_bev_pair(p1[0], p1[1], p2[0], p2[1], R, t, surface, mode="exact")
        bevs.append(r1["bev"])
        tiles += [bo.tile_from_bev(r1["bev"]), bo.tile_from_bev(r2["bev"])]
    x = th.from_numpy(np.concatenate(tiles, 0))
    with th.no_grad():
        ref = ro.forward(model.state_dict(), 152, [c[None] for c in x.split(3)])
    return bevs, ref[0].numpy()


def test_config5_launch_shape_against_the_oracle():
    """BASELINE config 5 at the shape `bench.py` runs it (its `config5` object; `bench.py --pano-hw 1024x2048 --surfaces floor,ceiling
    --layers 152 --panos 16`): 4096 hypotheses over 16 panoramas of 2048 x 1024, the launch size PICKED by the pipeline (pick_launch: the whole
    shard = 8192 renders per rasteriser launch -- a splat grid of 131 k workgroups, the densify stage in costly-first order with its tile
    phase writing the 16-channel samples), ResNet-152 with 12 input channels at batch 4096.  8 sampled hypotheses end to end against the
    oracle: both posed BEV images bit for bit, logits absolute 1e-3."""
    from salve_amd.models.early_fusion import EarlyFusionCEResnet
    from salve_amd.pipeline import RenderVerifyPipeline
    from _helpers import randomise_bn

    H, W, N, P = 1024, 2048, 4096, 16
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = EarlyFusionCEResnet(152, False, 2, SimpleNamespace(modalities=["ceiling_rgb_texture", "floor_rgb_texture"])).eval()
    randomise_bn(model)
    hyp = synthetic.make_hypotheses(N, P, seed=0)
    picked = np.random.default_rng(15).choice(N, 8, replace=False)
    with mp.get_context("spawn").Pool(8) as pool:
        pending = pool.map_async(_oracle_config5_hypothesis, [(hyp.i1[j], hyp.i2[j], hyp.R[j], hyp.t[j]) for j in picked], chunksize=1)
        panos = synthetic.make_panos(P, H, W)
        pipe = RenderVerifyPipeline(model, dev, pano_hw=(H, W), chunk=None, overlap=False, streams=1, n_hypotheses=N)   # bench.py's config5_line
        assert pipe.chunk == N, f"the pipeline was meant to pick the whole shard as one launch, picked {pipe.chunk}"
        pipe.load_panos(np.stack([p[0] for p in panos]), np.stack([p[1] for p in panos]))
        prepared = pipe.prepare(hyp)
        logits = pipe.score(prepared)
        torch.cuda.synchronize()
        pipe.check("config 5 launch shape")
        assert pipe.valid_mask(prepared).all() and torch.isfinite(logits).all()
        got_logits = logits.cpu().numpy()
        got_bev = {}
        for j in picked:
            ck, k0 = pipe.bev_index(prepared, int(j))
            got_bev[int(j)] = pipe.ras.export_u8(pipe.bevs[pipe.last_chunk_buffer[ck]][k0:k0 + 2]).cpu().numpy()   # ceiling, floor
        oracle = pending.get(timeout=1500)
    worst = 0.0
    for j, (bevs_exp, logit_exp) in zip(picked, oracle):
        for si, surface in enumerate(("ceiling", "floor")):
            assert np.array_equal(got_bev[int(j)][si], bevs_exp[si]), f"hypothesis {j}, {surface}: BEV image differs from the oracle"
        err = float(np.abs(got_logits[j] - logit_exp).max())
        worst = max(worst, err)
        assert err <= 1e-3, f"hypothesis {j}: logits {got_logits[j]} vs oracle {logit_exp}"
    print(f"config 5 launch shape ({N} hypotheses = {2 * N} renders per launch, ResNet-152 12-ch batch {N}): 8 hypotheses, "
          f"BEV bit-exact, max |logit - oracle| {worst:.2e}")
