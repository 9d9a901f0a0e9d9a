"""Kernel R's compile-time geometry (cleanrl_amd/csrc/convr_geom.h) checked on the host: tests/host/convr_geom_check.cpp is compiled with g++
against the header the device code includes and verifies, for the four instances the library launches, that every row of an image group sits in
exactly one lane slot, that the sixteen-lane sets of a fragment read touch sixteen different bank slots (no conflicts: the layer-2 forward had
23 per round of fragment reads before its rows were padded to 21 records -- 39 % of its LDS cycles, profiles/r05_pmc_lds.csv), that the k-step
orders are permutations, that (window origin) + (tap) is additive in the record index (the stride-2 layer stores even columns first), and the
LDS budgets."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_kernel_r_row_tables_and_offsets_hold_on_the_host(tmp_path):
    exe = str(tmp_path / "convr_geom_check")
    subprocess.run(["g++", "-std=c++20", "-O1", "-I" + os.path.join(ROOT, "cleanrl_amd", "csrc"), os.path.join(ROOT, "tests", "host", "convr_geom_check.cpp"),
                    "-o", exe], check=True, capture_output=True, text=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = {j["instance"]: j for j in map(json.loads, r.stdout.strip().splitlines())}
    assert set(got) == {"RConv2", "RConv3", "RDgrad3", "RDgrad2"}
    assert {k: (v["rows"], v["slots"], v["ksteps"]) for k, v in got.items()} == {
        "RConv2": (162, 192, 32), "RConv3": (245, 256, 36), "RDgrad3": (243, 256, 36), "RDgrad2": (200, 256, 16)}
    assert all(v["conflicts"] == 0 and v["lds_bytes"] <= 160 * 1024 for v in got.values()), got


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_kernel_rb_border_class_tiles_hold_on_the_host(tmp_path):
    """Kernel RB (cleanrl_amd/csrc/convrb_geom.h, the layer-2 data gradient by border class): every row of a three-image group sits in exactly one lane
    slot; a rim tile skips a ring slot (= a tap) only if that tap reads the zero border for ALL its rows -- the products it leaves out are exact zeros
    in kernel R / Z; image pixels and border records do not share LDS records although padded lines share their border record; the interior tiles'
    fragment reads are conflict-free; 32 tile-slot visits per group (128 tile-k-steps for three images; kernel R: 128 for two)."""
    exe = str(tmp_path / "convrb_geom_check")
    subprocess.run(["g++", "-std=c++20", "-O1", "-I" + os.path.join(ROOT, "cleanrl_amd", "csrc"), os.path.join(ROOT, "tests", "host", "convrb_geom_check.cpp"),
                    "-o", exe], check=True, capture_output=True, text=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = json.loads(r.stdout.strip())
    assert (got["rows"], got["slots"], got["tile_slot_visits_per_group"]) == (300, 320, 32)
    assert got["conflicts"] <= 4 and got["lds_bytes"] <= 160 * 1024 and got["rounds_per_thread"] == 8
