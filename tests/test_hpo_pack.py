"""SURVEY.md 8f row f4, host logic of the packed hyper-parameter sweep: round-robin shares cover every
configuration exactly once and the gathered result list is in configuration order on every rank
(3 gloo ranks on CPU; the trainings themselves need the GPU: tests/test_gpu_api.py::test_hpo_pack_*)."""
import os
import socket

import pytest
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_packed_indices_partition():
    from contrastiveprosthetics_amd import dist as cpdist
    for n in (0, 1, 7, 150):
        for w in (1, 2, 3, 8, 16):
            shares = [cpdist.packed_indices(n, r, w) for r in range(w)]
            assert sorted(i for s in shares for i in s) == list(range(n))
            assert max(len(s) for s in shares) - min(len(s) for s in shares) <= 1
    assert cpdist.gather_packed({0: "a", 1: "b"}, 2) == ["a", "b"]          # one process: nothing to gather
    with pytest.raises(RuntimeError):
        cpdist.gather_packed({0: "a"}, 2)


def _worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from contrastiveprosthetics_amd import dist as cpdist
    r, w, dev = cpdist.init_packed_from_env()
    assert (r, w, dev) == (rank, world, 0)
    n = 10
    mine = cpdist.packed_indices(n, r, w)
    got = cpdist.gather_packed({i: (float(i) * 0.5, r) for i in mine}, n)
    assert [g[0] for g in got] == [i * 0.5 for i in range(n)]
    assert [g[1] for g in got] == [i % world for i in range(n)]
    cpdist.shutdown()


@pytest.mark.timeout(120)
def test_gather_packed_three_ranks():
    mp.spawn(_worker, args=(3, _free_port()), nprocs=3, join=True)
