"""bench.py's own launcher (`python bench.py --gpus N` without torch.distributed.run): N children, one rendezvous,
the worst exit code, and no rank left waiting when another one dies. CPU only: the children are a stand-in script
that joins a gloo group instead of running the bench."""
import os
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _child(tmp_path, body):
    path = tmp_path / "child.py"
    path.write_text(textwrap.dedent(body))
    return str(path)


def test_launch_ranks_rendezvous_and_worst_exit_code(tmp_path):
    import bench
    out = tmp_path / "out"
    out.mkdir()
    script = _child(tmp_path, f"""
        import os, sys
        import torch, torch.distributed as dist
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["LOCAL_RANK"] == os.environ["RANK"] and os.environ["MASTER_ADDR"] == "127.0.0.1"
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([rank + 1])
        dist.all_reduce(t)
        open(os.path.join({str(out)!r}, f"rank{{rank}}"), "w").write(f"{{int(t)}} {{os.environ['MASTER_PORT']}} {{sys.argv[1:]}}")
        dist.destroy_process_group()
        sys.exit(3 if rank == 1 else 0)
    """)
    code = bench.launch_ranks(3, script=script, argv=["--steps", "2"])
    assert code == 3                                         # the worst of the children's exit codes
    seen = sorted(os.listdir(out))
    assert seen == ["rank0", "rank1", "rank2"]
    rows = [open(out / r).read().split(" ", 2) for r in seen]
    assert {r[0] for r in rows} == {"6"}                      # 1 + 2 + 3: they met in ONE group
    assert len({r[1] for r in rows}) == 1                     # ... on one port
    assert all(r[2] == "['--steps', '2']" for r in rows)     # the bench's own arguments are passed through


def test_launch_ranks_ends_the_others_when_a_rank_dies(tmp_path):
    import time
    import bench
    script = _child(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "0":
            sys.exit(7)
        time.sleep(600)            # a rank waiting for a peer that is gone
    """)
    t0 = time.time()
    code = bench.launch_ranks(2, script=script, argv=[])
    assert code in (7, 15)          # rank 0's code, or the SIGTERM the survivor was ended with
    assert time.time() - t0 < 60
