"""Two ring ranks (two processes, gloo transport, both on the one test GPU) driving real engine windows: the staggered
multi-sequence piped ring must generate exactly the tokens a single full-model window generates for each sequence."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _hp(d):
    return dict(arch=d.arch, n_layer=d.n_layer, n_embd=d.n_embd, n_head=d.n_head, n_head_kv=d.n_head_kv,
                head_dim=d.head_dim, n_ff=d.n_ff, n_vocab=d.n_vocab, rms_eps=d.rms_eps, rope_freq_base=d.rope_freq_base)


def _worker(rank, world, port, n_rounds, first, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from _bind import tiny_model
    import prima_cpp_amd.engine as E
    from prima_cpp_amd.ring import EngineCompute, RingDriver
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = tiny_model(np.random.default_rng(61), arch=0, n_layer=4, n_embd=256, n_head=4, n_head_kv=2, n_ff=512, n_vocab=320, n_ctx=64, rope_freqs=True)
    lo, hi = (0, 2) if rank == 0 else (2, 4)
    with torch.cuda.stream(torch.cuda.Stream()):
        w = E.Window(_hp(d), lo=lo, hi=hi, flags=(E.HAS_EMBD | E.HAS_HEAD) if rank == 0 else 0, n_ctx=64)
        w.load_desc(d)
        w.finalize(max_tokens=1, n_seq=world)
        comp = EngineCompute(w, world, use_graph=True)
        drv = RingDriver(comp, rank, world)
        hist = [[] for _ in range(world)]
        total = world * (n_rounds + 1)
        for m in range(total):
            forced = first[m] if (rank == 0 and m < world) else None
            seq = drv.micro_step(forced_token=forced)
            if rank == 0 and m >= world and seq is not None:
                torch.cuda.synchronize()
                hist[seq].append(int(comp.cur.item()))          # token that was just generated for `seq` and fed back
        drv.flush()
        torch.cuda.synchronize()
        if rank == 0:
            q.put(hist)
        w.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_ring_matches_single_window():
    import torch
    import torch.multiprocessing as mp
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    sys.path.insert(0, ROOT)
    from _bind import tiny_model
    import prima_cpp_amd.engine as E
    world, n_rounds = 2, 6
    first = [17, 101]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_rounds, first, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    # single full window, one sequence at a time, device greedy loop
    d = tiny_model(np.random.default_rng(61), arch=0, n_layer=4, n_embd=256, n_head=4, n_head_kv=2, n_ff=512, n_vocab=320, n_ctx=64, rope_freqs=True)
    w = E.Window(_hp(d), n_ctx=64)
    w.load_desc(d)
    w.finalize(1)
    for s in range(world):
        w.kv_clear()
        io = torch.zeros(n_rounds + 1, dtype=torch.int32, device="cuda")
        io[0] = first[s]
        w.generate(io, 0, n_rounds, use_graph=True)
        torch.cuda.synchronize()
        assert got[s][:n_rounds] == io.cpu().numpy()[1:].tolist(), (s, got[s], io.cpu().numpy())
    w.close()


def _worker_stag(rank, world, port, n_rounds, first, q, depth=1):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from _bind import tiny_model
    import prima_cpp_amd.engine as E
    from prima_cpp_amd.ring import CRing
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = tiny_model(np.random.default_rng(61), arch=0, n_layer=4, n_embd=256, n_head=4, n_head_kv=2, n_ff=512, n_vocab=320, n_ctx=64, rope_freqs=True)
    lo, hi = (0, 2) if rank == 0 else (2, 4)
    with torch.cuda.stream(torch.cuda.Stream()):
        w = E.Window(_hp(d), lo=lo, hi=hi, flags=(E.HAS_EMBD | E.HAS_HEAD) if rank == 0 else 0, n_ctx=64)
        w.load_desc(d)
        n_seq = depth * world                                                  # depth 2: two sequences per rank in flight (the two-deep schedule)
        w.finalize(max_tokens=1, n_seq=n_seq)
        ring = CRing(rank, world, transport="torch")
        total = n_seq * (n_rounds + 1)
        out = torch.full((total,), -1, dtype=torch.int32, device="cuda") if rank == 0 else None
        # in two calls: the schedule's state (micro-step counter, buffer toggles) lives in the ring object
        n1 = n_seq + 3
        ring.decode_staggered(w, n1, forced=(list(first) + [None] * (n1 - n_seq)) if rank == 0 else None, tokens_out=out, reset=True)
        ring.decode_staggered(w, total - n1, tokens_out=out[n1:] if rank == 0 else None)
        ring.wait()
        torch.cuda.synchronize()
        if rank == 0:
            fed = out.cpu().numpy().tolist()                                   # token fed at micro-step m = generated for sequence m % n_seq
            q.put([[fed[m] for m in range(n_seq + s_, total, n_seq)] for s_ in range(n_seq)])
        ring.close()
        w.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("depth", [1, 2])
def test_c_staggered_decode_loop_two_ranks_matches_single_window(depth):
    """pm355_ring_decode_staggered - the N-sequences-in-flight decode loop in C (was RingDriver.micro_step) - on two ranks sharing the test GPU over
    gloo (transport callbacks): every sequence's token stream equals the one a single full-model window generates for it. And the same loop
    on a world-1 local ring against Window.generate. depth 2 (round 6): windows finalized for 2 x world sequences run the two-deep schedule - a hop
    sent after micro-step m is consumed at m + 2, the exchange of step m travels under step m + 1's compute."""
    import torch
    import torch.multiprocessing as mp
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    sys.path.insert(0, ROOT)
    from _bind import tiny_model
    import prima_cpp_amd.engine as E
    from prima_cpp_amd.ring import CRing
    world, n_rounds = 2, 6
    first = [17, 101, 5, 230][:depth * world]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_stag, args=(r, world, port, n_rounds, first, q, depth)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    d = tiny_model(np.random.default_rng(61), arch=0, n_layer=4, n_embd=256, n_head=4, n_head_kv=2, n_ff=512, n_vocab=320, n_ctx=64, rope_freqs=True)
    w = E.Window(_hp(d), n_ctx=64)
    w.load_desc(d)
    w.finalize(1)
    want = []
    for s in range(depth * world):
        w.kv_clear()
        io = torch.zeros(n_rounds + 1, dtype=torch.int32, device="cuda")
        io[0] = first[s]
        w.generate(io, 0, n_rounds, use_graph=True)
        torch.cuda.synchronize()
        want.append(io.cpu().numpy()[1:].tolist())
        assert got[s][:n_rounds] == want[s], (s, got[s], want[s])
    # world 1, no communicator: forced first token, then the head's argmax fed back
    w.kv_clear()
    w.set_pos(0)
    ring = CRing(0, 1, transport="local")
    out = torch.full((n_rounds + 1,), -1, dtype=torch.int32, device="cuda")
    ring.decode_staggered(w, n_rounds + 1, forced=[first[0]] + [None] * n_rounds, tokens_out=out, reset=True)
    ring.wait()
    torch.cuda.synchronize()
    assert out.cpu().numpy()[1:].tolist() == want[0], (out.cpu().numpy(), want[0])
    ring.close()
    w.close()


def test_c_ring_transport_world1_self_send():
    """The RCCL transport in C (pm355_ring_*, prima_cpp_amd/csrc/ring.hip) with world size 1: rank 0's next and previous rank are
    itself, so one grouped ncclSend + ncclRecv moves a buffer through the communicator; then whole micro-steps through
    pm355_ring_step (wait -> window step -> exchange) against plain engine steps. Multi-rank RCCL needs one GPU per rank (the
    driver's --gpus N run); the staggered schedule itself is covered by the gloo tests."""
    import ctypes as C
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    sys.path.insert(0, ROOT)
    from _bind import tiny_model
    import prima_cpp_amd.engine as E
    from prima_cpp_amd.ring import CRing
    E.torch = torch
    ring = CRing(0, 1)
    a = torch.randn(1, 4096, device="cuda")
    b = torch.zeros_like(a)
    ring.exchange(a, b)
    ring.wait()
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    # micro-steps: x_out of step i is sent to "the next rank" (= this rank) and arrives as the input buffer of step i + 1
    rng = np.random.default_rng(11)
    d = tiny_model(rng, arch=0, n_layer=2, n_embd=256, n_head=4, n_head_kv=2, n_ff=512, n_vocab=320, n_ctx=64, rope_freqs=True)
    hp = dict(arch=d.arch, n_layer=d.n_layer, n_embd=d.n_embd, n_head=d.n_head, n_head_kv=d.n_head_kv, head_dim=d.head_dim,
              n_ff=d.n_ff, n_vocab=d.n_vocab, rms_eps=d.rms_eps, rope_freq_base=d.rope_freq_base)
    outs = []
    for use_ring in (False, True):
        w = E.Window(hp, n_ctx=64)
        w.load_desc(d)
        w.finalize(max_tokens=1)
        w.set_pos(0)
        tok = torch.tensor([5], dtype=torch.int32, device="cuda")
        x_out = [torch.zeros(1, d.n_embd, device="cuda") for _ in range(2)]
        x_in = [torch.zeros(1, d.n_embd, device="cuda") for _ in range(2)]
        am = torch.zeros(1, dtype=torch.int32, device="cuda")
        lib = w.lib
        lib.pm355_ring_step.restype = C.c_int
        lib.pm355_ring_step.argtypes = [C.c_void_p] * 7 + [C.c_int] * 5 + [C.c_void_p, C.c_int64, C.c_void_p]
        res = []
        for i in range(6):
            st = torch.cuda.current_stream().cuda_stream
            if use_ring:
                rc = lib.pm355_ring_step(ring.h, w.h, tok.data_ptr(), None, x_out[i & 1].data_ptr(), None, am.data_ptr(), 1, 0, 0, 1,
                                         1, x_in[(i + 1) & 1].data_ptr(), d.n_embd, st)
                assert rc == 0, lib.pm355_ring_error()
                ring.wait()
                torch.cuda.synchronize()
                assert torch.equal(x_in[(i + 1) & 1], x_out[i & 1])          # what was sent is what arrived
            else:
                w.step(token=tok, x_out=x_out[i & 1], argmax=am, advance=1)
                torch.cuda.synchronize()
            res.append((x_out[i & 1].cpu().numpy().copy(), int(am.item())))
            tok.copy_(am)
        outs.append(res)
        w.close()
    for (h0, t0), (h1, t1) in zip(*outs):
        assert t0 == t1 and np.array_equal(h0, h1)
    ring.close()


def _worker_c(rank, world, port, prompts, ubatch, n_single, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from _bind import tiny_model
    import prima_cpp_amd.engine as E
    from prima_cpp_amd.ring import CRing
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = tiny_model(np.random.default_rng(61), arch=0, n_layer=4, n_embd=256, n_head=4, n_head_kv=2, n_ff=512, n_vocab=320, n_ctx=64, rope_freqs=True)
    lo, hi = (0, 2) if rank == 0 else (2, 4)
    n_seq, n_prompt = prompts.shape
    with torch.cuda.stream(torch.cuda.Stream()):
        w = E.Window(_hp(d), lo=lo, hi=hi, flags=(E.HAS_EMBD | E.HAS_HEAD) if rank == 0 else 0, n_ctx=64)
        w.load_desc(d)
        w.finalize(max_tokens=ubatch, n_seq=n_seq)
        ring = CRing(rank, world, transport="torch")             # the C ring over gloo: both ranks on the one GPU
        toks = torch.from_numpy(prompts.astype(np.int32)).cuda() if rank == 0 else None
        rows = torch.zeros((n_seq, d.n_embd), dtype=torch.float32, device="cuda") if rank == 0 else None
        ring.prefill(w, toks, n_seq, n_prompt, ubatch, rows)     # pipelined prompt pass, [n_tokens][n_embd] per hop
        torch.cuda.synchronize()
        out = {}
        if rank == 0:
            out["rows"] = rows.cpu().numpy()
            first = []
            lg = torch.empty(d.n_vocab, dtype=torch.float32, device="cuda")
            am = torch.empty(1, dtype=torch.int32, device="cuda")
            for s in range(n_seq):
                w.head(rows[s], logits=lg, argmax=am)
                torch.cuda.synchronize()
                first.append(int(am.item()))
            out["first"] = first
        # ONE sequence in flight (the reference's mode): sequence 0, n_single tokens once round the ring each
        tok = torch.tensor([out["first"][0]], dtype=torch.int32, device="cuda") if rank == 0 else None
        gen = []
        for _ in range(n_single):
            ring.single_token(w, 0, tok)
            if rank == 0:
                torch.cuda.synchronize()
                gen.append(int(tok.item()))
        ring.wait()
        torch.cuda.synchronize()
        if rank == 0:
            out["single"] = gen
            q.put(out)
        ring.close()
        w.close()
    dist.barrier()
    dist.destroy_process_group()


def test_c_ring_pipelined_prefill_and_single_stream_two_ranks():
    """pm355_ring_prefill (ubatch-wide hand-off [n_tokens][n_embd], chunk g on rank r at pipeline step g + r, the last rank returns each
    prompt's last row to rank 0) and pm355_ring_single_token (one sequence in flight, the reference's mode) - the C schedule on two ranks
    sharing the test GPU over gloo (transport callbacks) - against ONE full-model window fed the same chunks."""
    import torch
    import torch.multiprocessing as mp
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    sys.path.insert(0, ROOT)
    from _bind import tiny_model
    import prima_cpp_amd.engine as E
    world, ubatch, n_single = 2, 4, 5
    prompts = np.random.default_rng(9).integers(0, 320, (2, 11))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_c, args=(r, world, port, prompts, ubatch, n_single, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    d = tiny_model(np.random.default_rng(61), arch=0, n_layer=4, n_embd=256, n_head=4, n_head_kv=2, n_ff=512, n_vocab=320, n_ctx=64, rope_freqs=True)
    for s in range(prompts.shape[0]):
        w = E.Window(_hp(d), n_ctx=64)
        w.load_desc(d)
        w.finalize(max_tokens=ubatch)
        hid = None
        for c0 in range(0, prompts.shape[1], ubatch):
            chunk = torch.from_numpy(prompts[s, c0:c0 + ubatch].astype(np.int32)).cuda()
            hid, lg, _ = w.decode(tokens=chunk, pos0=c0)
        torch.cuda.synchronize()
        assert np.array_equal(hid[-1].cpu().numpy(), got["rows"][s]), s          # same kernels, same order: bit-identical
        first = int(np.argmax(lg.cpu().numpy()))
        assert first == got["first"][s]
        if s == 0:
            io = torch.zeros(n_single + 1, dtype=torch.int32, device="cuda")
            io[0] = first
            w.generate(io, prompts.shape[1], n_single, use_graph=True)
            torch.cuda.synchronize()
            assert got["single"] == io.cpu().numpy()[1:].tolist(), (got["single"], io.cpu().numpy())
        w.close()


def test_bench_two_ranks_on_one_gpu_end_to_end():
    """`bench.py --gpus 2` exactly as the driver launches it (torch.distributed.run, one process per rank), with PM355_DIST_BACKEND=gloo so
    that both ranks can share the one test GPU: window partition, the C ring (pipelined prompt pass with [n_tokens][n_embd] hand-offs,
    staggered multi-sequence decode, single-sequence loop) and the JSON contract. The first real multi-GPU RCCL run is then not the first
    run of this code."""
    import json
    import subprocess
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    env = dict(os.environ, PM355_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--model", "llama3-8b", "--steps", "4", "--warmup", "2",
           "--no-extras", "--no-cpu-baseline", "--prefill", "0", "--n-ctx", "2048"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["metric"] == "decode_tokens_per_s" and d["value"] > 0 and d["scaling"] == "weak"
    assert "pp2" in d["config"]["parallelism"] and "ring in C" in d["config"]["parallelism"]
    assert d["single_stream"]["tokens_per_s"] > 0 and d["ring_prefill"]["tokens_per_s"] > 0 and d["ring_prefill"]["ubatch"] == 512
    print(f"\n[bench --gpus 2, gloo, one GPU] aggregate {d['value']:.1f} tok/s, single stream {d['single_stream']['tokens_per_s']:.1f} tok/s, "
          f"ring prompt pass {d['ring_prefill']['tokens_per_s']:.0f} tok/s")
