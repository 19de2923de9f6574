"""CPU-only tests of bench.py's measurement helpers (no GPU, no rocprofv3): the PMC counter parser behind `roofline.traffic`, and the recorded fall-back."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_counter_parser_picks_the_render_kernel_rows(tmp_path):
    p = tmp_path / "t_counter_collection.csv"
    p.write_text(
        '"Correlation_Id","Dispatch_Id","Agent_Id","Queue_Id","Process_Id","Thread_Id","Grid_Size","Kernel_Id","Kernel_Name","Workgroup_Size","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Counter_Name","Counter_Value","Start_Timestamp","End_Timestamp"\n'
        '1,1,"Agent 2",1,1,1,512,10,"__amd_rocclr_fillBufferAligned",256,0,0,8,0,48,"FETCH_SIZE",4.75,1,2\n'
        '2,2,"Agent 2",1,1,1,262144,11,"void nrs::render_kernel_c128<8, false, false, false, 0, 0>(nrs::DeviceModel, nrs::RenderArgs)",512,70304,0,64,0,112,"FETCH_SIZE",3000000.0,3,4\n'
        '3,3,"Agent 2",1,1,1,262144,11,"void nrs::render_kernel_c128<8, false, false, false, 0, 0>(nrs::DeviceModel, nrs::RenderArgs)",512,70304,0,64,0,112,"FETCH_SIZE",3200000.0,5,6\n'
        '4,4,"Agent 2",1,1,1,262144,11,"void nrs::render_kernel_c128<8, false, false, false, 0, 0>(nrs::DeviceModel, nrs::RenderArgs)",512,70304,0,64,0,112,"WRITE_SIZE",31000.0,7,8\n')
    assert bench.render_kernel_counter_values([str(p)], "FETCH_SIZE") == [3000000.0, 3200000.0]
    assert bench.render_kernel_counter_values([str(p)], "WRITE_SIZE") == [31000.0]
    assert bench.render_kernel_counter_values([str(p)], "TCC_MISS_sum") == []
    assert bench.render_kernel_counter_values([], "FETCH_SIZE") == []


def test_recorded_traffic_is_there_for_the_workloads_that_name_it():
    j = json.load(open(os.path.join(ROOT, bench.TRAFFIC_FILE)))
    assert j["traffic_bytes_per_launch"] == int(2 * j["fetch_size_kb"] * 1024 + j["write_size_kb"] * 1024)   # the guide's gfx950 correction, applied once
    t, src = bench.measured_traffic("lego_cage")
    assert t == j["traffic_bytes_per_launch"] and "recorded" in src
    t2, src2 = bench.measured_traffic("garden_cage_records64")   # (round 6: the 64 GiB configuration is the one with a committed PMC pass)
    assert t2 == j["garden_cage_records64"]["traffic_bytes_per_launch"] and "recorded" in src2
    t3, src3 = bench.measured_traffic("lego_cage_varied")   # (round 5: the varied-opacity scene has a committed PMC pass too)
    assert t3 == j["lego_cage_varied"]["traffic_bytes_per_launch"] and "recorded" in src3
    assert bench.measured_traffic("lego_cage_norecords") == (None, None)   # a workload without a committed pass: null, never a guess


def test_live_traffic_can_be_switched_off(monkeypatch):
    monkeypatch.setenv("NRS_BENCH_LIVE_TRAFFIC", "0")
    t, why = bench.live_traffic("lego_cage", 1920, 1080)
    assert t is None and "switched off" in why
