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
