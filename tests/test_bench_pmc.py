"""CPU tests of the bench's counter passes (bench_pmc.py): the CSV parsing and the arithmetic behind `roofline.traffic` and
`*.sustained_clock_ghz` / `*.frac_at_sustained_clock`, on canned rocprofv3 counter_collection files - no GPU, no profiler."""

import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench_pmc  # noqa: E402

HEADER = ('"Correlation_Id","Dispatch_Id","Agent_Id","Queue_Id","Process_Id","Thread_Id","Grid_Size","Kernel_Id","Kernel_Name",'
          '"Workgroup_Size","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Counter_Name","Counter_Value",'
          '"Start_Timestamp","End_Timestamp"\n')
K3 = "void vk::vk_theory_cells_kernel<3, 3, 0, 0, 0>(vk::TheoryArgs)"
KB = "void vk::vk_theory_cells_kernel<1, 2, 0, 0, 0>(vk::TheoryArgs)"
KK = "void vk::vk_theory_cells_kernel<1, 2, 0, 4, 0>(vk::TheoryArgs)"
KL = "void vk::vk_like_tiled_kernel<8>(vk::LikeArgs)"


def write_csv(path, rows):
    """rows: (dispatch id, kernel, counter, value, start, end)"""
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as fh:
        fh.write(HEADER)
        for d, k, c, v, t0, t1 in rows:
            fh.write(f'{d},{d},"Agent 2",1,77,77,16777216,41,"{k}",256,0,0,64,0,96,"{c}",{v:.6f},{t0},{t1}\n')


def clock_rows(ghz_by_kernel, plan, t=1_000_000):
    """Dispatch records of a clock pass: plan = [(kernel, launches, duration ns)], an image and a likelihood kernel in between."""
    rows, d = [], 1
    rows.append((d, "void vk::vk_image_kernel<3>(vk::TheoryArgs, int, int, double*, int)", "GRBM_GUI_ACTIVE", 8e4, t, t + 10_000))
    d += 1
    for kernel, launches, dur in plan:
        for _ in range(launches):
            t += 50_000
            rows.append((d, kernel, "GRBM_GUI_ACTIVE", 8.0 * ghz_by_kernel[kernel] * dur, t, t + dur))
            d += 1
            t += dur
            rows.append((d, KL, "GRBM_GUI_ACTIVE", 8.0 * 2.0 * 150_000, t, t + 150_000))
            d += 1
            t += 150_000
    return rows


def test_sustained_clock_from_a_canned_counter_file(tmp_path):
    ghz = {K3: 1.98, KB: 2.11, KK: 2.00}
    plan = [(K3, 8, 22_640_000), (KB, 8, 3_460_000), (KK, 6, 441_000), (KK, 6, 430_000)]
    write_csv(str(tmp_path / "x" / "123_counter_collection.csv"), clock_rows(ghz, plan))
    rows = bench_pmc.read_counter_rows(str(tmp_path), "GRBM_GUI_ACTIVE")
    assert len(rows) == 28 and all("vk_theory" in r["kernel"] for r in rows)          # image and likelihood kernels are not counted
    assert [r["dispatch"] for r in rows] == sorted(r["dispatch"] for r in rows)
    seq = [{"label": "config3", "warm": 2, "timed": 6, "event_ms": 22.7}, {"label": "boss_cmass", "warm": 2, "timed": 6},
           {"label": "kaiser", "warm": 2, "timed": 4}, {"label": "euclid_special", "warm": 2, "timed": 4}]
    clocks = bench_pmc.clocks_from_rows(rows, seq)
    assert clocks["config3"]["sustained_clock_ghz"] == pytest.approx(1.98, rel=1e-9)
    assert clocks["boss_cmass"]["sustained_clock_ghz"] == pytest.approx(2.11, rel=1e-9)
    # kaiser and euclid_special run the SAME instantiation: only the order of the dispatches tells them apart
    assert clocks["kaiser"]["kernel"] == clocks["euclid_special"]["kernel"] == "vk::vk_theory_cells_kernel<1, 2, 0, 4, 0>"
    assert clocks["kaiser"]["dispatch_ms"] == pytest.approx(0.441) and clocks["euclid_special"]["dispatch_ms"] == pytest.approx(0.430)
    assert clocks["config3"]["dispatches"] == 6 and clocks["config3"]["child_event_ms"] == 22.7
    # what the line does with it: the fraction against the peak at that clock
    import bench
    flops = 15_424_000 * 65536
    f = bench.clock_fields(flops, clocks, "config3")
    assert f["clock_source"] == "this run" and f["sustained_clock_ghz"] == pytest.approx(1.98)
    assert clocks["config3"]["cycles_per_dispatch"] == pytest.approx(1.98 * 22_640_000)
    # cycle domain: flops of a launch / its shader cycles / the peak's flops per cycle ...
    assert f["frac_at_sustained_clock"] == pytest.approx(flops / (1.98 * 22_640_000) / (78.6e12 / 2.4e9))
    # ... which is `frac` x 2.4 GHz / clock when the time is that of the same dispatches (22.64 ms: 0.568)
    frac = flops / 22.64e-3 / 78.6e12
    assert frac == pytest.approx(0.568, abs=1e-3) and f["frac_at_sustained_clock"] == pytest.approx(bench.at_sustained_clock(frac, 1.98))
    # no pass, or a workload the pass did not cover: null, never another run's figure
    for none in (bench.clock_fields(flops, None, "config3"), bench.clock_fields(flops, clocks, "dispersion")):
        assert none == {"sustained_clock_ghz": None, "frac_at_sustained_clock": None, "clock_source": None}


def test_a_clock_is_never_guessed(tmp_path):
    ghz = {K3: 1.98, KB: 2.11}
    write_csv(str(tmp_path / "pmc_counter_collection.csv"), clock_rows(ghz, [(K3, 8, 22_640_000), (KB, 8, 3_460_000)]))
    rows = bench_pmc.read_counter_rows(str(tmp_path), "GRBM_GUI_ACTIVE")
    ok = [{"label": "config3", "warm": 2, "timed": 6}, {"label": "boss_cmass", "warm": 2, "timed": 6}]
    assert bench_pmc.clocks_from_rows(rows, ok) is not None
    # the child launched more (or fewer) kernels than the profiler recorded: no clocks at all
    assert bench_pmc.clocks_from_rows(rows, ok + [{"label": "kaiser", "warm": 2, "timed": 4}]) is None
    assert bench_pmc.clocks_from_rows(rows[:-1], ok) is None
    assert bench_pmc.clocks_from_rows([], []) is None
    # a workload whose timed dispatches are of two kernels (the sequence is out of step with the records)
    bad = [{"label": "config3", "warm": 2, "timed": 8}, {"label": "boss_cmass", "warm": 0, "timed": 6}]
    assert bench_pmc.clocks_from_rows(rows, bad) is None
    # records without timestamps
    for r in rows:
        r["end_ns"] = None
    assert bench_pmc.clocks_from_rows(rows, ok) is None


def test_traffic_from_canned_counter_files(tmp_path):
    # config 3, 65536 points: FETCH_SIZE 5322 KiB (x 2 on gfx950), WRITE_SIZE 61440 KiB per K1 launch; K2 reads the workspace back
    f_rows = [(i, K3 if i % 2 else KL, "FETCH_SIZE", 5322.0 if i % 2 else 31405.0, 10 * i, 10 * i + 5) for i in range(1, 9)]
    w_rows = [(i, K3 if i % 2 else KL, "WRITE_SIZE", 61440.0 if i % 2 else 1024.0, 10 * i, 10 * i + 5) for i in range(1, 9)]
    write_csv(str(tmp_path / "f" / "pmc_counter_collection.csv"), f_rows)
    write_csv(str(tmp_path / "w" / "pmc_counter_collection.csv"), w_rows)
    fr = bench_pmc.read_counter_rows(str(tmp_path / "f"), "FETCH_SIZE")
    wr = bench_pmc.read_counter_rows(str(tmp_path / "w"), "WRITE_SIZE")
    assert len(fr) == 4 and len(wr) == 4                                              # the likelihood kernel is not the theory kernel
    t = bench_pmc.traffic_from_rows(fr, wr)
    assert t["read_bytes"] == 2 * 5322 * 1024 and t["written_bytes"] == 61440 * 1024
    assert t["bytes_per_launch"] == t["read_bytes"] + t["written_bytes"] and t["launches_averaged"] == 4
    assert t["kernel"] == "vk::vk_theory_cells_kernel<3, 3, 0, 0, 0>"


def test_the_clock_child_record_is_found_in_its_stdout():
    seq = [{"label": "config3", "warm": 2, "timed": 6, "event_ms": 22.6}]
    out = "rocprofv3 banner\n{not json\n" + json.dumps({"something": 1}) + "\n" + json.dumps({"clock_pass": seq}) + "\ntrailing text\n"
    assert bench_pmc.parse_clock_child(out) == seq
    assert bench_pmc.parse_clock_child("no record here\n") is None


def test_the_bench_quotes_no_clock_of_another_run():
    """`frac_at_sustained_clock` must come from this run's own pass: bench.py has no reader of a stored clock any more."""
    with open(os.path.join(ROOT, "bench.py")) as fh:
        src = fh.read()
    assert "profiled_clock" not in src and "effective_clock_ghz" not in src
    assert "clock_fields(" in src and "bench_pmc.live_clocks(" in src
