"""CPU tier: `python bench.py --gpus N` without an external launcher becomes the parent of N ranks - it must compose the
torch.distributed.run command (rendezvous on 127.0.0.1, one process per GPU, its own arguments passed through) and must
not have imported torch or touched HIP when it does so (never re-exec / fork a process that initialised the GPU)."""
import json
import os
import subprocess
import sys
import textwrap

from testlib import REPO


def test_parent_composes_the_rank_launch_without_touching_the_gpu():
    script = textwrap.dedent(f"""
        import json, subprocess, sys
        sys.argv = ["bench.py", "--gpus", "4", "--steps", "7", "--rehearse-gloo"]
        sys.path.insert(0, {REPO!r})
        seen = {{}}
        def fake_call(cmd, env=None):
            seen["cmd"], seen["legacy_ipc"], seen["torch_loaded"] = cmd, env.get("HSA_ENABLE_IPC_MODE_LEGACY"), "torch" in sys.modules
            return 0
        subprocess.call = fake_call
        import bench
        try:
            bench.main()
        except SystemExit as exc:
            seen["exit"] = exc.code
        print(json.dumps(seen))
    """)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    seen = json.loads(res.stdout.strip().splitlines()[-1])
    cmd = seen["cmd"]
    assert seen["exit"] == 0 and seen["torch_loaded"] is False and seen["legacy_ipc"] == "0"
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    tail = cmd[cmd.index(os.path.join(REPO, "bench.py")) + 1:]
    assert tail == ["--gpus", "4", "--steps", "7", "--rehearse-gloo"]


def test_a_rank_does_not_launch_again():
    """under a launcher (RANK set) main() must go on to the rank path, not spawn ranks of its own"""
    script = textwrap.dedent(f"""
        import subprocess, sys
        sys.argv = ["bench.py", "--gpus", "2"]
        sys.path.insert(0, {REPO!r})
        def boom(*a, **k):
            raise AssertionError("a rank tried to launch ranks")
        subprocess.call = boom
        import bench
        bench.launch_ranks = boom
        try:
            bench.main()
        except AssertionError:
            raise
        except BaseException as exc:      # no GPU here: the rank path fails later, which is fine
            print("rank path:", type(exc).__name__)
    """)
    env = dict(os.environ, RANK="1", WORLD_SIZE="2", LOCAL_RANK="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    res = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=300)
    assert "a rank tried to launch ranks" not in res.stderr and "rank path:" in res.stdout, res.stdout + res.stderr


def test_kernel_source_hash_ignores_comments_and_layout_only():
    """roofline.traffic is reported only when profiles/hbm_traffic.json was captured from this build's kernel CODE: the hash
    must not move when a comment is reworded, and must move when a token changes."""
    sys.path.insert(0, REPO)
    import bench
    a = 'int f(int x) { /* old words */ return x + 1; }  // tail\nconst char *s = "// kept /* kept */";\n'
    b = 'int f(int x) {\n    return x + 1;   /* new\n words */\n}\nconst char *s = "// kept /* kept */";'
    c = a.replace("x + 1", "x + 2")
    assert bench._strip_c_comments(a) == bench._strip_c_comments(b) != bench._strip_c_comments(c)
    assert '"// kept /* kept */"' in bench._strip_c_comments(a)
    h = bench.kernel_source_sha()
    assert len(h) == 64 and h == bench.kernel_source_sha()


def test_exit_rule_follows_the_oracle_not_zero():
    """at delta = 4 or n = 63 the reference itself loses bits; the bench must fail only when the GPU's
    errors differ from the oracle's on the sample both ran (or, without a sample, when a provably error-free setting loses one)"""
    import bench
    ok = {"gpu_round_trip_bit_errors_on_sample": 6584, "oracle_round_trip_bit_errors_on_sample": 6584,
          "gpu_extract_of_reference_stego_bit_mismatches": 0}
    assert bench.round_trip_verdict(ok, 3891153, 4.0, 3) is None                 # the reference's own 1.6 % BER, reproduced
    assert bench.round_trip_verdict(dict(ok, gpu_round_trip_bit_errors_on_sample=6585), 0, 8.0, 3)
    assert bench.round_trip_verdict(dict(ok, gpu_extract_of_reference_stego_bit_mismatches=1), 0, 8.0, 3)
    assert bench.round_trip_verdict(None, 0, 8.0, 3) is None
    assert bench.round_trip_verdict(None, 5, 8.0, 3)                             # provably error-free setting lost bits
    assert bench.round_trip_verdict(None, 5, 8.0, 63) is None                    # clipping: the reference loses bits too
    assert bench.round_trip_verdict(None, 5, 4.0, 3) is None
    assert bench.round_trip_verdict(None, 5, 64.0, 3) is None                    # large steps clip


def test_a_wrong_gather_cannot_exit_zero_and_a_correct_one_cannot_fail_at_any_delta():
    """the collective's exit rule is the senders' digests against what rank 0 received - fatal when they
    differ, silent about the payload (at delta = 4 the reference itself loses 1.67 % of the bits)"""
    import torch

    import bench
    ok = {"gpu_round_trip_bit_errors_on_sample": 0, "oracle_round_trip_bit_errors_on_sample": 0,
          "gpu_extract_of_reference_stego_bit_mismatches": 0}
    for delta in (4.0, 8.0, 16.0, 64.0):
        assert bench.round_trip_verdict(None, 12345, delta, 3, gather_ok=False)          # a wrong gather is fatal everywhere
        assert bench.round_trip_verdict(ok, 0, delta, 3, gather_ok=False)
    assert bench.round_trip_verdict(None, 3891153, 4.0, 3, gather_ok=True) is None       # configs[4]'s delta = 4 leg: correct run passes
    assert bench.round_trip_verdict(None, 0, 8.0, 3, gather_ok=True) is None
    assert bench.round_trip_verdict(None, 0, 8.0, 3, gather_ok=None) is None
    # the digest: equal for equal bytes, different when two slices change hands or a piece moves inside a slice
    g = torch.Generator().manual_seed(7)
    a = torch.randint(0, 256, (1003,), dtype=torch.uint8, generator=g)
    b = torch.randint(0, 256, (1003,), dtype=torch.uint8, generator=g)
    da, db = bench.stream_digest(a, 1000), bench.stream_digest(b, 1000)
    assert torch.equal(da, bench.stream_digest(a.clone(), 1000)) and not torch.equal(da, db)
    moved = a.clone()
    moved[:8], moved[8:16] = a[8:16], a[:8]                                              # same multiset of words, other order
    assert not torch.equal(da, bench.stream_digest(moved, 1000))
    beyond = a.clone()
    beyond[1001] ^= 0xff                                                                 # bytes past n_bytes do not count
    assert torch.equal(da, bench.stream_digest(beyond, 1000))
    flipped = a.clone()
    flipped[999] ^= 1
    assert not torch.equal(da, bench.stream_digest(flipped, 1000))


def test_roofline_kernel_label_follows_the_launch_predicate():
    """the label is derived from the predicate of csrc/svs_capi.hip, delta range included"""
    import bench
    assert bench.embed_kernel_label("guarded", 3, 8.0).startswith("embed_row1_kernel")
    assert bench.embed_kernel_label("guarded", 10, 8.0).startswith("embed_kernel<2>")
    for mode, n, delta in (("exact", 3, 8.0), ("guarded", 16, 8.0), ("guarded", 3, 0.1), ("guarded", 3, 8192.0), ("guarded", 63, 8.0)):
        assert bench.embed_kernel_label(mode, n, delta).startswith("embed_exact_kernel"), (mode, n, delta)


def test_documented_multi_gpu_commands_parse_and_plan():
    """the multi-GPU commands of the BASELINE configurations as README.md documents them are parsed by
    bench.py's own parser and dry-run through bench.job_plan (pure arithmetic, no GPU): contiguous shares that cover the clip
    in stream order, each rank's bit offset, equal gather contributions, one call's block count inside the ABI limit, a rank's
    buffers inside one MI355X's HBM - and the exit rule cannot fail a correct run of any of them (no CPU sample at N > 1)."""
    import re
    import shlex
    sys.path.insert(0, REPO)
    import bench
    from svsdct import batch
    text = open(os.path.join(REPO, "README.md")).read()
    cmds = [shlex.split(line)[2:] for line in re.findall(r"^python bench\.py --gpus .*$", text, flags=re.M)]
    assert len(cmds) == 5
    seen = set()
    for argv in cmds:
        args = bench.parse_args(argv)
        world = args.gpus
        assert world == 8
        plan = bench.job_plan(args, world)
        seen.add((args.total_frames, args.height, args.width, args.n_ac, args.delta))
        per_frame = batch.capacity_bits(1, args.height, args.width, args.n_ac)
        assert plan["per_frame_bits"] == per_frame
        nxt = 0
        for r, rk in enumerate(plan["ranks"]):
            if args.total_frames:
                assert rk["first_frame"] == nxt                          # contiguous, rank order = stream order
            nxt = rk["first_frame"] + rk["frames"]
            assert rk["frames"] > 0 and rk["first_bit"] == rk["first_frame"] * per_frame
            assert rk["blocks"] < 2 ** 31                                # one svs_embed_dev call per rank and step (include/svsdct.h)
            assert rk["device_bytes"] < 0.5 * 288e9                      # fits one MI355X with room to spare
            assert (rk["capacity_bits"] + 7) // 8 <= plan["gather_bytes_per_rank"]
        if args.total_frames:
            assert nxt == args.total_frames == plan["total_frames"] and plan["scaling"] == "strong"
        else:
            assert plan["total_frames"] == world * args.frames and plan["scaling"] == "weak"
        assert plan["bytes_received_by_rank0_per_step"] == world * plan["gather_bytes_per_rank"]
        # N > 1 has no oracle sample: the run may only fail on a setting that provably loses no bit; the reference's own
        # payload errors (delta = 4: 1.6 %; n = 10 on clipping content) never fail a correct run
        assert bench.round_trip_verdict(None, 0, args.delta, args.n_ac) is None
        if args.delta < 8 or args.n_ac > 7:
            assert bench.round_trip_verdict(None, 12345, args.delta, args.n_ac) is None
    assert (2400, 1080, 1920, 10, 8.0) in seen                            # configs[3]
    assert {(1200, 4320, 7680, 3, d) for d in (4.0, 8.0, 16.0)} <= seen   # configs[4]
    assert (0, 2160, 3840, 3, 8.0) in seen                                # configs[2] x N, the driver's scaling run
    # configs[3] shares: 300 frames per GPU = one 1080p clip each; configs[4]: 150 8K frames per GPU, 29.16 MB of bits each
    p3 = bench.job_plan(bench.parse_args(cmds[1]), 8)
    assert [c for _, c in p3["shares"]] == [300] * 8 and p3["gather_bytes_per_rank"] == 300 * 32400 * 10 // 8
    p4 = bench.job_plan(bench.parse_args(cmds[2]), 8)
    assert [c for _, c in p4["shares"]] == [150] * 8 and p4["gather_bytes_per_rank"] == 150 * 518400 * 3 // 8
    # uneven division and more ranks than some share sizes still cover the clip
    odd = bench.job_plan(bench.parse_args(["--gpus", "8", "--total-frames", "1203", "--height", "1080", "--width", "1920"]), 8)
    assert [c for _, c in odd["shares"]] == [151] * 3 + [150] * 5 and sum(c for _, c in odd["shares"]) == 1203
