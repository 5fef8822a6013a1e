"""bench.py --gpus N starts N ranks itself (the reference's launch(main, num_gpus, ...), tools/train_net.py:140-147):
exercised on CPU with the gloo backend and a stub workload."""
import json
import os
import sys

import bench


def _run(capfd, monkeypatch, argv):
    monkeypatch.setenv("YOLO_BENCH_WORKLOAD", "tests.bench_stub:make")
    monkeypatch.setenv("YOLO_BENCH_BACKEND", "gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("PYTHONPATH", os.pathsep.join([bench.ROOT] + sys.path))
    rc = bench.main(argv)
    out = capfd.readouterr().out
    lines = [l for l in out.splitlines() if l.startswith("{")]
    return rc, lines


def test_gpus_2_starts_two_ranks(capfd, monkeypatch):
    rc, lines = _run(capfd, monkeypatch, ["--gpus", "2", "--steps", "3", "--warmup", "1", "--min-seconds", "0.1"])
    assert rc == 0
    assert len(lines) == 1, lines  # rank 0 alone prints
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1
    assert len(rec["per_rank_value"]) == 2
    # rank 1's steps take twice as long: the job's value is set by the slowest rank (MAX over ranks)
    assert rec["per_rank_value"][0] > rec["per_rank_value"][1] * 1.3
    assert abs(rec["value"] - 2 * rec["per_rank_value"][1]) / rec["value"] < 0.15
    assert rec["timed_steps"] % 3 == 0 and rec["timed_steps"] >= 3
    assert rec["config"]["steps_run_on_rank0"] == 1 + 3 + rec["timed_steps"]
    assert "cpu_baseline" not in rec  # rank 0 at N = 1 only
    # the training block is measured at N > 1 too, by all ranks together (its all-reduce saw both of them)
    assert rec["train"]["n_gpus"] == 2 and rec["train"]["params_in_sync"] and rec["train"]["exchange"] == "allreduce"
    assert rec["n_ranks_gloo"] == 2
    # every rank names the device it ran on
    assert rec["per_rank_device"] == ["stub device 0", "stub device 1"] and rec["ranks_on_distinct_devices"]


def test_gpus_1_is_one_process(capfd, monkeypatch):
    rc, lines = _run(capfd, monkeypatch, ["--steps", "2", "--warmup", "0", "--min-seconds", "0"])
    assert rc == 0 and len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["timed_steps"] == 2 and len(rec["per_rank_value"]) == 1
    assert rec["cpu_baseline"]["kind"] == "port"
    assert rec["train"]["n_gpus"] == 1


def test_no_train_flag_skips_the_training_block(capfd, monkeypatch):
    rc, lines = _run(capfd, monkeypatch, ["--gpus", "2", "--steps", "1", "--warmup", "0", "--min-seconds", "0", "--no-train"])
    assert rc == 0 and len(lines) == 1
    assert "train" not in json.loads(lines[0])


def test_a_failing_rank_fails_the_launch(monkeypatch, capfd):
    monkeypatch.setenv("YOLO_BENCH_WORKLOAD", "tests.bench_stub:no_such_factory")
    monkeypatch.setenv("YOLO_BENCH_BACKEND", "gloo")
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setenv("PYTHONPATH", os.pathsep.join([bench.ROOT] + sys.path))
    assert bench.main(["--gpus", "2", "--steps", "1", "--warmup", "0"]) != 0
    capfd.readouterr()


def test_a_training_block_that_hangs_does_not_take_the_headline_line_with_it(capfd, monkeypatch):
    """At world > 1 the training block is a collective; if it never completes (a rank stuck, a transport problem), the
    watchdog prints the line measured so far with the reason in place of the training record, and the job ends."""
    monkeypatch.setenv("YOLO_BENCH_WORKLOAD", "tests.bench_stub:make_hanging")
    monkeypatch.setenv("YOLO_BENCH_BACKEND", "gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("PYTHONPATH", os.pathsep.join([bench.ROOT] + sys.path))
    rc = bench.main(["--gpus", "2", "--steps", "1", "--warmup", "0", "--min-seconds", "0", "--train-timeout", "3"])
    lines = [l for l in capfd.readouterr().out.splitlines() if l.startswith("{")]
    assert rc == 0 and len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and "watchdog" in rec["train"]["error"]


def test_ranks_sharing_a_device_show_in_the_line(capfd, monkeypatch):
    monkeypatch.setenv("YOLO_BENCH_WORKLOAD", "tests.bench_stub:make_same_device")
    monkeypatch.setenv("YOLO_BENCH_BACKEND", "gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("PYTHONPATH", os.pathsep.join([bench.ROOT] + sys.path))
    rc = bench.main(["--gpus", "2", "--steps", "1", "--warmup", "0", "--min-seconds", "0", "--no-train"])
    lines = [l for l in capfd.readouterr().out.splitlines() if l.startswith("{")]
    assert rc == 0 and len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["per_rank_device"] == ["stub device 0"] * 2 and rec["ranks_on_distinct_devices"] is False
