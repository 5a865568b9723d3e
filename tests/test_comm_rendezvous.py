"""The collective path's host logic at world size > 1, on the CPU: csrc/smj_comm.h -- the code smj_comm_init / smj_allgather_returns /
smj_comm_destroy consist of apart from hipSetDevice -- compiled into tests/rccl_stub/comm_harness and run as N processes against a stub
librccl (tests/rccl_stub/rccl_stub.cpp, bound through SMJ_RCCL_LIB like a site's own RCCL build would be).  What this exercises before
an 8-GPU box is the first to: dlopen + the five symbols, rank 0 publishing the ncclUniqueId (temporary + rename), the other ranks
waiting for it, the job nonce refusing a stale file of another job, the time-out messages, ncclCommInitRank / ncclAllGather /
ncclCommDestroy being called with the ABI smj_comm.h spells out, and the rank-major layout of the gathered returns."""
import os
import struct
import subprocess
import tempfile

import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_stub")


@pytest.fixture(scope="module")
def harness():
    subprocess.check_call(["make", "-s", "-C", HERE])
    return os.path.join(HERE, "comm_harness"), os.path.join(HERE, "librccl_stub.so")


def launch(harness, rank, world, path, timeout, count, env_extra, rounds=1):
    exe, lib = harness
    env = dict(os.environ, SMJ_RCCL_LIB=lib, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", WORLD_SIZE=str(world))
    env.pop("SMJ_JOB_NONCE", None)
    env.update(env_extra)
    return subprocess.Popen([exe, str(rank), str(world), path, str(timeout), str(count), str(rounds)], env=env, stdout=subprocess.PIPE, text=True)


def values(line):
    assert line.startswith("OK"), line
    return [float(v) for v in line.split()[1:]]


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_meet_through_the_id_file_and_gather_rank_major(harness, world):
    count = 5
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "smj_rccl_id")
        procs = [launch(harness, r, world, path, 20, count, {"SMJ_STUB_DIR": d}, rounds=3) for r in reversed(range(world))]   # rank 0 starts LAST
        outs = [p.communicate(timeout=60)[0].strip() for p in procs]
        assert all(p.returncode == 0 for p in procs), outs
        want = [1000.0 * r + i + 0.5 for r in range(world) for i in range(count)]      # third round: + 0.25 * 2
        for o in outs:
            assert values(o) == want
        assert not os.path.exists(path + ".tmp")


def test_a_stale_id_file_of_another_job_is_not_accepted(harness):
    """Right magic, right size, another job's nonce: rank 1 keeps waiting; once rank 0 of THIS job publishes, it joins."""
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "smj_rccl_id")
        with open(path, "wb") as f:
            f.write(b"SMJRCCL1" + struct.pack("<Q", 12345) + b"stub-deadbeef".ljust(128, b"\0"))
        p1 = launch(harness, 1, 2, path, 0.6, 3, {"SMJ_STUB_DIR": d})
        out = p1.communicate(timeout=30)[0]
        assert p1.returncode == 1 and out.startswith("ERR -7") and "belongs to another job" in out, out
        # the same stale file, and this job's rank 0 arriving a little later: replaced, both ranks through
        p1 = launch(harness, 1, 2, path, 20, 3, {"SMJ_STUB_DIR": d})
        p0 = launch(harness, 0, 2, path, 20, 3, {"SMJ_STUB_DIR": d})
        o1, o0 = p1.communicate(timeout=60)[0], p0.communicate(timeout=60)[0]
        assert p0.returncode == 0 and p1.returncode == 0, (o0, o1)
        assert values(o0) == values(o1) == [0.0, 1.0, 2.0, 1000.0, 1001.0, 1002.0]


def test_two_jobs_with_different_nonces_do_not_cross(harness):
    """SMJ_JOB_NONCE (or torchrun's MASTER_PORT / run id) separates jobs that were given the same id path."""
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "smj_rccl_id")
        a0 = launch(harness, 0, 2, path, 20, 2, {"SMJ_STUB_DIR": d, "SMJ_JOB_NONCE": "job-a"})
        b1 = launch(harness, 1, 2, path, 0.8, 2, {"SMJ_STUB_DIR": d, "SMJ_JOB_NONCE": "job-b"})
        ob = b1.communicate(timeout=30)[0]
        assert b1.returncode == 1 and "another job" in ob, ob
        a1 = launch(harness, 1, 2, path, 20, 2, {"SMJ_STUB_DIR": d, "SMJ_JOB_NONCE": "job-a"})
        assert a0.wait(timeout=60) == 0 and a1.wait(timeout=60) == 0


def test_missing_rank_0_times_out_with_the_path_in_the_message(harness):
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "never_written")
        p = launch(harness, 1, 2, path, 0.3, 2, {"SMJ_STUB_DIR": d})
        out = p.communicate(timeout=30)[0]
        assert p.returncode == 1 and out.startswith("ERR -7 timed out waiting for the RCCL id file") and path in out


def test_argument_errors_and_world_1(harness):
    with tempfile.TemporaryDirectory() as d:
        p = launch(harness, 2, 2, os.path.join(d, "x"), 1, 2, {"SMJ_STUB_DIR": d})
        assert "ERR -1 bad rank 2 / world 2" in p.communicate(timeout=30)[0]
        p = launch(harness, 1, 2, "", 1, 2, {"SMJ_STUB_DIR": d})
        assert "id_path is required" in p.communicate(timeout=30)[0]
        p = launch(harness, 0, 1, "", 1, 3, {"SMJ_STUB_DIR": d})       # world 1 through the communicator path: no file at all
        assert values(p.communicate(timeout=30)[0]) == [0.0, 1.0, 2.0] and os.listdir(d) != ["x"]
        exe, _ = harness
        env = dict(os.environ, SMJ_RCCL_LIB="/nonexistent/librccl.so")
        out = subprocess.run([exe, "0", "1", "", "1", "2"], env=env, capture_output=True, text=True).stdout
        assert out.startswith("ERR -7 SMJ_RCCL_LIB")
