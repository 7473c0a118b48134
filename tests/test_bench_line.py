"""CPU: bench.py's output contract.  The driver parses ONE stdout line; round 5's 21 KB line came back `parsed: null`, so the
final line is capped (< 4 KB) and everything else goes to gpurun_out/bench_detail.json.  Also: `bench.py --gpus 8` without torchrun
must reach the single-process multi-GPU path with device ids 0..7 (the first 8-GPU run must not die in argument handling)."""
import importlib.util
import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", ROOT / "bench.py")
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b


RECORDED = sorted(p.name for p in (ROOT / "profiles").glob("r0[56]_bench_bn254_2p20*.json"))


@pytest.mark.parametrize("name", RECORDED)
def test_compact_line_of_a_recorded_result(name):
    b = load_bench()
    full = json.load(open(ROOT / "profiles" / name))
    full["detail"] = b.DETAIL_PATH
    txt = b.compact_line(full)
    assert len(txt) < 4096 and "\n" not in txt
    line = json.loads(txt)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "detail"):
        assert k in line, k
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"]
    assert line["config"]["workload"].startswith("Groth16 prove")
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(line["roofline"])
    assert line["roofline"]["frac"] == full["roofline"]["frac"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"])
    assert line["msm_g1"]["mops"] == full["msm_g1"]["mops"]
    assert line["summary"]["bn254_2p20"]["v"] == full["value"]


def test_compact_line_never_exceeds_the_cap_and_keeps_the_contract_keys():
    b = load_bench()
    fat = {"metric": "m", "value": 1.0, "unit": "proofs/s", "n_gpus": 8, "steps": 2, "warmup": 1, "ms_per_step": 3.0,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "x" * 5000,
           "config": {"workload": "w" * 5000, "key_form": "k" * 5000, "parallelism": "p" * 5000},
           "roofline": {"bound": "hbm", "achieved": 1.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.1, "traffic": None, "kernel": "a" * 999,
                        "note": "n" * 9999},
           "cpu_baseline": {"value": 1, "unit": "proofs/s", "cores": 2, "kind": "port", "sample": "s" * 9999},
           "summary": {"blob": "z" * 6000}, "exchange": {"kind": "e" * 999}, "detail": "d"}
    txt = b.compact_line(fat)
    line = json.loads(txt)
    assert len(txt) < b.LINE_CAP
    assert "summary" not in line and line["roofline"]["frac"] == 0.1 and line["cpu_baseline"]["cores"] == 2
    assert line["n_gpus"] == 8


def test_multi_gpu_lines_quote_the_n1_cpu_baseline():
    b = load_bench()
    blk = b.recorded_cpu_baseline("bn254", 20)
    assert blk is not None and blk["kind"] == "port" and "from" in blk and blk["from"].startswith("profiles/")
    assert b.recorded_cpu_baseline("bn254", 13) is None


def test_emit_writes_the_detail_file_and_one_line(tmp_path, monkeypatch, capsys):
    b = load_bench()
    monkeypatch.chdir(tmp_path)
    full = json.load(open(ROOT / "profiles" / RECORDED[0]))
    b.emit(full)
    out = capsys.readouterr().out
    assert out.count("\n") == 1
    assert json.loads(out)["detail"] == b.DETAIL_PATH
    assert json.load(open(tmp_path / b.DETAIL_PATH))["value"] == full["value"]


class _Stop(Exception):
    pass


@pytest.mark.parametrize("argv, want_ids, want_mode", [
    (["--gpus", "8", "--steps", "2", "--warmup", "1"], list(range(8)), "throughput"),
    (["--gpus", "8", "--mode", "shard", "--log-n", "24", "--steps", "2", "--warmup", "1"], list(range(8)), "shard"),
    (["--devices", "0,0,0"], [0, 0, 0], "throughput")])
def test_gpus_8_without_torchrun_reaches_the_in_library_multi_gpu_path(monkeypatch, argv, want_ids, want_mode):
    """no GPU here: MultiContext is replaced by a recorder; main() must hand it ids 0..7 before anything touches a device"""
    b = load_bench()
    import ckb_zkp_amd.api as api
    seen = {}

    class FakeMulti:
        def __init__(self, ids):
            seen["ids"] = list(ids)
            raise _Stop()

    monkeypatch.setattr(api, "MultiContext", FakeMulti)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
    orig = b.bench_single_process_multi

    def spy(args):
        seen["mode"] = args.mode
        return orig(args)

    monkeypatch.setattr(b, "bench_single_process_multi", spy)
    with pytest.raises(_Stop):
        b.main()
    assert seen["ids"] == want_ids and seen["mode"] == want_mode
