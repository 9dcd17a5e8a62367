"""BASELINE configs 2 / 3 at their full size on the GPU: 4096 hypotheses over 64 synthetic 1024x512 panoramas through the
fused pipeline exactly as bench.py runs it (chunks of 1024, three HIP streams).

* 32 randomly chosen hypotheses are checked bit for bit against the oracle: the BEV pixel index of every panorama point
  (row a4, the bit-exact index contract) and the final BEV image (rows a5 - a8) -- taken from the buffers the overlapped
  run itself produced, not from a separate quiet render;
* all 4096 logits and the BEV images still resident are compared between the three-stream and the one-stream schedule:
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
            results[streams] = (logits.clone(), bev_last, pipe)
        oracle = pending.get(timeout=900)

    l3, b3, pipe3 = results[3]
    l1, b1, _ = results[1]
    assert torch.isfinite(l3).all()
    assert torch.equal(l3, l1), "logits differ between the three-stream and the one-stream schedule"
    assert torch.equal(b3[3], b1[3]), "BEV images of the last chunk differ between the schedules"

    # bit-exact against the oracle, from the overlapped run's own buffers
    ras = pipe3.ras
    for j, (bev_exp, xy_exp) in zip(picked, oracle):
        chunk, slot = divmod(int(j), CHUNK)
        got = ras.export_u8(b3[chunk][slot:slot + 1]).cpu().numpy()[0]
        assert bev_exp is not None
        assert np.array_equal(got, bev_exp), f"hypothesis {j}: BEV image differs from the oracle"
        # indices: re-render this hypothesis alone with the debug outputs
        row = ras.upload_hypotheses(pack_hypotheses([hyp.i1[j]], [0], hyp.R[j:j + 1], hyp.t[j:j + 1], [1]))
        _, dbg = ras.render(pipe3.pano_rgb, pipe3.pano_depth, row, 1, debug=True)
        xy = dbg.img_xy[0].cpu().numpy()
        assert np.array_equal(xy[xy[:, 0] >= 0], xy_exp), f"hypothesis {j}: pixel indices differ"
        assert int(dbg.in_window[0]) == xy_exp.shape[0]
