"""bench.py's host logic that needs no GPU: the self-launch of the N ranks for a plain `python bench.py --gpus N`
(the driver's command shape), and the PMC traffic lookup that withholds figures taken on other kernel sources."""
import importlib
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def bench(monkeypatch):
    monkeypatch.syspath_prepend(ROOT)
    return importlib.import_module("bench")


def test_plain_gpus_n_starts_a_child_torchrun_before_any_gpu_call(bench, monkeypatch):
    import subprocess

    import torch

    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env

        class R:
            returncode = 7

        return R()

    def no_gpu(*a, **k):
        raise AssertionError("the parent process touched the GPU before launching its ranks")

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(torch.cuda, "set_device", no_gpu)
    monkeypatch.setattr(bench._lib, "load", no_gpu)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2", "--config", "4"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7  # the child's exit code is ours
    cmd = seen["cmd"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1 :] == ["--gpus", "4", "--steps", "7", "--warmup", "2", "--config", "4"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_under_torchrun_no_second_launch(bench, monkeypatch):
    """WORLD_SIZE set (the driver's torchrun form): the rank runs itself; a mismatch with --gpus is an error."""
    monkeypatch.setattr(bench, "self_launch", lambda n: (_ for _ in ()).throw(AssertionError("launched again")))
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "does not match WORLD_SIZE" in str(e.value.code)


def test_pmc_traffic_is_withheld_for_other_sources(bench, monkeypatch, tmp_path):
    from mathlib_amd.build import source_hash

    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    body = {"source_hash": "0" * 16, "commit": "abc", "configs": {"2": {"k_accumulate28_seg<Bls381>": {"FETCH_SIZE": 10.0, "WRITE_SIZE": 6.0}}}}
    (prof / "r03_pmc_traffic.json").write_text(json.dumps(body))
    t, note = bench._pmc(2, "k_accumulate28_seg<Bls381")
    assert t is None and "withheld" in note
    body["source_hash"] = source_hash()
    (prof / "r03_pmc_traffic.json").write_text(json.dumps(body))
    t, note = bench._pmc(2, "k_accumulate28_seg<Bls381")
    assert t == 16.0 * 1024 and "abc" in note
    assert bench._pmc(5, "k_accumulate28_seg<Bls377")[0] is None
