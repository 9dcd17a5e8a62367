"""SURVEY section 8f row 3, host side: the hypothesis work list of a floor (file naming of
scripts/export_alignment_hypotheses.py:234-238, enumeration of scripts/render_dataset_bev.py:80-110, Sim(2) JSON of
salve/common/sim2.py:180-188) and the names of the tiles it leads to."""

import json
import shutil
from pathlib import Path

import numpy as np

from salve_amd import ingest
from salve_amd.common.sim2 import Sim2

GOLDEN = Path(__file__).resolve().parent / "golden"


def write_sim2(path: Path, theta_deg: float, t, s=1.0):
    th = np.deg2rad(theta_deg)
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    path.parent.mkdir(parents=True, exist_ok=True)
    with open(path, "w") as f:   # save_Sim2: {"R": flat 4, "t": flat 2, "s": scale}, indent 4 (export_alignment_hypotheses.py:75-90)
        json.dump({"R": R.flatten().tolist(), "t": list(t), "s": s}, f, indent=4)
    return R


def test_reference_sim2_fixture_loads():
    S = Sim2.from_json(GOLDEN / "a_Sim2_b.json")     # the reference's own tests/test_data/a_Sim2_b.json
    assert np.array_equal(S.rotation, np.eye(2, dtype=np.float32)) and S.rotation.dtype == np.float32
    assert np.array_equal(S.translation, np.array([3930, 3240], dtype=np.float32)) and np.isclose(S.scale, 5 / 3)


def test_floor_work_list_order_and_names(tmp_path):
    root = tmp_path / "hyp"
    floor = root / "0715" / "floor_02"
    R_a = write_sim2(floor / "gt_alignment_approx" / "4_38__door_3_0_identity.json", 30.0, (0.25, -1.5))
    write_sim2(floor / "gt_alignment_approx" / "12_7__opening_0_1_rotated.json", 200.0, (1.0, 2.0))
    write_sim2(floor / "incorrect_alignment" / "5_6__window_10_2_identity.json", 90.0, (0.0, 0.0), 1.01)
    write_sim2(floor / "incorrect_alignment" / "38_4__door_0_3_rotated.json", 181.0, (-2.0, 0.5))
    shutil.copy(GOLDEN / "a_Sim2_b.json", floor / "incorrect_alignment" / "9_10__door_1_1_identity.json")
    write_sim2(root / "0715" / "floor_01" / "gt_alignment_approx" / "1_2__door_0_0_identity.json", 0.0, (0, 0))  # another floor

    h = ingest.load_floor_hypotheses(str(root), "0715", "floor_02")
    assert len(h) == 5
    # labels in the order (gt_alignment_approx, incorrect_alignment); inside a label sorted by path; pair_idx restarts
    assert [Path(p).name for p in h.fpaths] == ["12_7__opening_0_1_rotated.json", "4_38__door_3_0_identity.json",
                                                "38_4__door_0_3_rotated.json", "5_6__window_10_2_identity.json", "9_10__door_1_1_identity.json"]
    assert h.label.tolist() == [1, 1, 0, 0, 0] and h.pair_idx.tolist() == [0, 1, 0, 1, 2]
    assert h.i1.tolist() == [12, 4, 38, 5, 9] and h.i2.tolist() == [7, 38, 4, 6, 10]
    assert h.pair_uuid == ["opening_0_1_rotated", "door_3_0_identity", "door_0_3_rotated", "window_10_2_identity", "door_1_1_identity"]
    assert h.R.dtype == np.float32 and h.t.dtype == np.float32                     # Sim2 stores float32 (sim2.py:50-52)
    assert np.array_equal(h.R[1], R_a.astype(np.float32)) and np.array_equal(h.t[1], np.array([0.25, -1.5], np.float32))
    assert np.array_equal(h.t[4], np.array([3930, 3240], np.float32)) and np.isclose(h.s[4], 5 / 3)

    panos = {i: f"/zind/0715/panos/floor_02_partial_room_{i % 7:02d}_pano_{i}.jpg" for i in (4, 5, 6, 7, 9, 10, 12, 38)}
    names = h.tile_names("/bev", panos)
    # (fp0, fp1) in FILE-NAME order, as the reference's dataset sorts a pair's tiles (zind_data.py:110): pano 38 lives in
    # partial_room_03, pano 4 in partial_room_04, so the tile of i2 = 38 comes first although i1 = 4
    assert names[1] == ("/bev/gt_alignment_approx/0715/pair_1___door_3_0_identity_floor_rgb_floor_02_partial_room_03_pano_38.jpg",
                        "/bev/gt_alignment_approx/0715/pair_1___door_3_0_identity_floor_rgb_floor_02_partial_room_04_pano_4.jpg")
    assert h.swap(panos).tolist() == [True, True, False, False, False]
    # the order is lexicographic on the stem, not numeric on the pano id: pano_10 < pano_9 inside one partial room
    same_room = {9: "/z/floor_02_partial_room_01_pano_9.jpg", 10: "/z/floor_02_partial_room_01_pano_10.jpg"}
    assert h.swap({**panos, **same_room}).tolist()[4] is True
    assert [Path(n).name.split("_pano_")[1] for n in h.tile_names("/bev", {**panos, **same_room})[4]] == ["10.jpg", "9.jpg"]
    assert names[2][0].startswith("/bev/incorrect_alignment/0715/pair_0___door_0_3_rotated_floor_rgb_")
    # the names parse back under the dataset rules
    from salve_amd.dataset import zind_data
    assert zind_data.pair_idx_from_fpath(names[4][0]) == 2 and zind_data.pano_id_from_fpath(names[4][1]) == 10
    assert ingest.panoid_from_fpath(panos[38]) == 38
    assert len(ingest.load_floor_hypotheses(str(root), "0715", "floor_04")) == 0
