"""The RCCL branch of the single-process multi-device entry points with MORE THAN ONE rank, without an 8-GPU node: a stand-in
for librccl.so (tests/fake_rccl.c, opened through TRM_RCCL_LIBRARY) implements group semantics and the all-reduce through host
memory for eight communicators in one process, on the box's one GPU.  Until the driver's 8-GPU run the grouped
ncclCommInitRank / ncclAllReduce sequences of trm_comm_init_all, trm_reduce_global_all and trm_status_global_all had only ever
executed with a world of one.  This validates SEQUENCING and PACKING only (which calls, in which order, inside which group;
the NaN flags that travel with min / max; the OR of the status bits) -- not transport, not scaling: no curve has been measured."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def result():
    env = dict(os.environ)
    for k in ("TRM_RCCL_LIBRARY", "TRM_RCCL_ALLOW_SHARED_DEVICE", "FAKE_RCCL_FAIL_INIT_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fake_rccl_worker.py")], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
    assert line, out.stdout[-2000:]
    return json.loads(line[-1][7:])


def test_the_rccl_branch_runs_with_eight_communicators(result):
    assert result["world"] == [8] * 8 and result["rank"] == list(range(8))       # trm_comm_info: the grouped init took, not the host fold
    assert result["live_after_init"] == 8


def test_grouped_reductions_equal_the_host_fold_bitwise(result):
    assert result["reductions_checked"] == 17 and result["reductions_equal_host_fold_bitwise"]
    assert result["status_clean"] == [0, 0]
    assert result["single_rank"]


def test_a_nan_in_one_shard_reaches_every_rank(result):
    assert result["nan_reaches_every_rank"]
    assert result["status_after_nan"][0] == result["status_after_nan"][1] and result["status_after_nan"][0] & 1


def test_lists_that_are_not_one_group_in_rank_order_are_refused_not_posted(result):
    assert result["refused"] == {k: "refused/refused" for k in ("subset", "mixed_groups", "out_of_order", "some_without")}
    assert result["group_c_works"]


def test_a_failing_rank_leaves_no_communicator_behind(result):
    assert result["init_failure"] == "refused"
    assert result["world_after_failure"] == [0] * 8
    assert result["world_after_retry"] == [8] * 8 and result["live_delta_retry"] == 8
    assert result["live_after_destroy"] == 0
