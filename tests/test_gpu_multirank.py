"""The N > 1 path with the REAL contexts: fresh child processes, one per rank, each owning the
context of its block of the N145 columns on the box's GPU; global diagnostics and the gathered output must equal the
single-context run.  Plus the library's own RCCL diagnostics (trm_comm_init / trm_reduce_global) with the one rank a
one-GPU box allows (RCCL refuses two ranks on one device; the multi-GPU run is the driver's)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import workloads as W
import terrarium_jl_amd as trm

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("config,world", [("richards", 2), ("land", 3)])
def test_block_sharded_contexts_in_fresh_processes(config, world):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multirank_worker.py"), config, "20"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank}:\n{out[-3000:]}"
    assert "multirank ok" in outs[0]


def test_library_communicator_single_rank():
    """trm_comm_init / trm_reduce_global / trm_status_global through RCCL with world size 1: the collective path of the
    library (dlopen of librccl, communicator, side stream, all-reduce) runs and returns the local values."""
    lat, lon = W.columns_from_mask("N72")
    w = W.make_workload("richards", lat[:500], lon[:500], 20)
    d = W.setup_device(w)
    d.step(w["dt"], 5, finalize=True)
    assert d.comm_world() == 0
    with pytest.raises(trm.TerrariumHipError):
        d.reduce_global("temperature", "sum")            # no communicator yet
    d.comm_init(0, 1, d.comm_unique_id())
    assert d.comm_world() == 1
    T = d.get("temperature")
    for op in ("sum", "min", "max", "hasnan"):
        assert np.array_equal(d.reduce_global("temperature", op), d.reduce("temperature", op)), op
    assert d.reduce_global("saturation_water_ice", "volume_integral_z")[0] == d.reduce("saturation_water_ice", "volume_integral_z")[0]
    T[4, 7] = np.nan
    d.set("temperature", T)
    assert np.isnan(d.reduce_global("temperature", "min")[4]) and np.isnan(d.reduce_global("temperature", "max")[4])
    assert d.reduce_global("temperature", "hasnan")[4] == 1
    assert d.status_global() == d.status()
    from terrarium_jl_amd import parallel
    assert np.array_equal(parallel.global_reduce(d, "internal_energy", "max"), d.reduce("internal_energy", "max"))
    d.close()
