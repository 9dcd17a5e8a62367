"""One-off wide parity sweep (GPU box): N random hypotheses x {floor, ceiling} over several panoramas, final BEV images
bit for bit against the oracle's exact mode, oracle renders in a process pool.  python tools/measure/parity_sweep.py [N] [procs] [scene] [seed] [HxW]
(HxW: the panoramas' size, default 512x1024; 1024x2048 is BASELINE config 5's -- four times the points per render, keep N <= 1536)"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import multiprocessing as mp
import numpy as np


def oracle_one(args):
    from oracle import bev_oracle as bo
    from salve_amd import synthetic
    pi, surface, R, t, scene, hw = args
    rgb, depth = synthetic.make_pano(pi, hw[0], hw[1], scene=scene)
    a = bo.xyzrgb_from_arrays(depth, rgb, bo.floor_ceiling_z_range(surface))
    a, _ = bo.pose_pair(a, a[:1], R, t)
    res = bo.render_bev_image(a, mode="exact")
    return None if res is None else res["bev"]


if __name__ == "__main__":
    import torch
    from salve_amd import synthetic
    from salve_amd.rasteriser import BevRasteriser, pack_hypotheses
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    scene = sys.argv[3] if len(sys.argv) > 3 else "box"
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 123
    hw = tuple(int(v) for v in sys.argv[5].lower().split("x")) if len(sys.argv) > 5 else (512, 1024)
    P = 6
    hyp = synthetic.make_hypotheses(N, P, seed=seed)
    hyp.t[::7] *= 2.0     # some clouds half out of the window
    surf = np.arange(N) % 2
    jobs = [(int(hyp.i1[j]), "floor" if surf[j] == 0 else "ceiling", hyp.R[j], hyp.t[j], scene, hw) for j in range(N)]
    t0 = time.time()
    with mp.get_context("spawn").Pool(procs) as pool:
        ref = pool.map(oracle_one, jobs, chunksize=2)
    print(f"oracle: {time.time() - t0:.0f} s for {N} renders", flush=True)
    dev = torch.device("cuda:0")
    ras = BevRasteriser(dev, pano_hw=hw)
    panos = [synthetic.make_pano(i, hw[0], hw[1], scene=scene) for i in range(P)]
    d_rgb, d_depth = ras.upload_panos(np.stack([p[0] for p in panos]), np.stack([p[1] for p in panos]))
    h = pack_hypotheses(hyp.i1, surf, hyp.R, hyp.t, np.ones(N))
    bad = 0
    # (r6) reps 1 and 2 run the pipeline's form: scatter, then the densify stage whose last phase writes the verifier tile (salve_bev_densify_tiles);
    # every 16th render's tile is compared with the oracle's Resize -> Crop -> Normalize of ITS image (fp16-rounded), the second image being zeros
    from oracle import bev_oracle as bo
    crop = ras.crop
    tiles_b = torch.zeros((1, crop, crop), dtype=torch.int32, device=dev)
    r_ = np.arange(N)
    jobs_a = ras.upload_tile_jobs(np.zeros(N, dtype=np.int64), r_, np.zeros(N, dtype=np.int64))
    jobs_b = ras.upload_tile_jobs(np.zeros(N, dtype=np.int64), r_, np.full(N, 3, dtype=np.int64), pretiled=True)
    tiles = torch.zeros((N, crop, crop, 8), dtype=torch.float16, device=dev)
    bad_tiles = 0
    for rep in range(3):   # repeated: the triangle cache and the work distribution are timing dependent
        hd = ras.upload_hypotheses(h)
        if rep == 0:
            bev = ras.render(d_rgb, d_depth, hd, N)[0]
        else:
            bev = torch.empty((N, *ras.bev_hw), dtype=torch.int32, device=dev)
            ras.scatter(d_rgb, d_depth, hd, N, bev)
            ras.densify_tiles(N, bev, jobs_a, jobs_b, tiles_b, tiles, 8)
            tg = tiles[::16, :, :, :3].float().cpu().numpy()
            for k, j in enumerate(range(0, N, 16)):
                exp = ref[j] if ref[j] is not None else np.zeros((501, 501, 3), np.uint8)
                want = bo.tile_from_bev(exp).astype(np.float16).astype(np.float32).transpose(1, 2, 0)
                if not np.array_equal(tg[k], want):
                    bad_tiles += 1
                    print("TILE MISMATCH rep", rep, "render", j, flush=True)
        got = ras.export_u8(bev).cpu().numpy()
        for j in range(N):
            exp = ref[j] if ref[j] is not None else np.zeros((501, 501, 3), np.uint8)
            if not np.array_equal(got[j], exp):
                bad += 1
                print("MISMATCH rep", rep, "render", j, jobs[j][:2], int((got[j] != exp).any(-1).sum()), "pixels", flush=True)
    from salve_amd import status
    status.check(dev, "parity_sweep")
    print(f"scene {scene}, panoramas {hw[1]}x{hw[0]}, seed {seed}: renders compared:", 3 * N, "mismatches:", bad, "; tiles of the fused phase compared:", 2 * len(range(0, N, 16)), "mismatches:", bad_tiles)
    sys.exit(1 if (bad or bad_tiles) else 0)
