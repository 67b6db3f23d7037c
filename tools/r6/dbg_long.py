"""round-6 debugging aid: where does an 8200-token prompt at the 70B head shape stall? Stages printed unbuffered, every stage under its own timeout."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import _fixtures8d as F
from _bind import Ref, run_llama_driver, best_ref_flavour
from test_gpu_parity_8d import SHAPES, GPU_ARGS
size = sys.argv[1] if len(sys.argv) > 1 else "70bh"
n_prompts = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "600,1100,8200").split(",")]
ref = Ref(best_ref_flavour())
plain, pk = "/tmp/s8d_dbg_plain.gguf", "/tmp/s8d_dbg_peaked.gguf"
t0 = time.time()
F.write_model(plain, ref, tag=size, **SHAPES[size]); F.copy_with_new_head(plain, pk, ref, True, tag=size)
print(f"gguf {time.time() - t0:.0f} s", flush=True)
V = SHAPES[size]["n_vocab"]
for n_prompt in n_prompts:
    prompt = F.prompt_tokens(V, n_prompt)
    for what in ("plugin", "engine", "cpu8", "cpu16"):
        t0 = time.time()
        try:
            if what == "plugin":
                tg, lg, st = run_llama_driver(pk, prompt, 4, ngl=99, n_ctx=8448, threads=8, timeout=240, extra_args=GPU_ARGS)
                print(f"prompt {n_prompt} plugin {time.time() - t0:.1f} s tokens {tg.tolist()}", flush=True)
            elif what.startswith("cpu"):
                tc, lc, sc = run_llama_driver(pk, prompt, 4, ngl=0, n_ctx=8448, threads=int(what[3:]), flavour=best_ref_flavour(), timeout=400)
                print(f"prompt {n_prompt} reference CPU {what[3:]} threads {time.time() - t0:.1f} s tokens {tc.tolist()}", flush=True)
            else:
                te, le = F.engine_greedy(pk, SHAPES[size], prompt, 4, 8448)
                print(f"prompt {n_prompt} engine {time.time() - t0:.1f} s tokens {te.tolist()}", flush=True)
        except Exception as e:
            print(f"prompt {n_prompt} {what} FAILED after {time.time() - t0:.1f} s: {str(e)[-600:]}", flush=True)
