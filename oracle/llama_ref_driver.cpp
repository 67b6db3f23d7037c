// llama_ref_driver.cpp — TEST INFRASTRUCTURE (our own code, links the reference's libllama + common built by
// oracle/Makefile from /root/reference; nothing under prima_cpp_amd/ uses it).
//
// Greedy decode through the reference's UNMODIFIED driver: gpt_params_parse (common/arg.cpp:323) ->
// llama_init_from_gpt_params (common/common.cpp:1663: load, layer window, backends, KV, schedulers, warm-up) ->
// llama_decode (src/llama.cpp:18229 llama_decode_internal). The command line is the reference's own
// (-m, -ngl, --keep-out-in-cuda, -c, -t, -fa, -ctk/-ctv ...); the harness-only knobs come from the environment so the
// reference's argument parser is used untouched:
//   REFDRV_PROMPT  comma-separated token ids (default: "1")   REFDRV_NGEN  tokens to generate (default 16)
//   REFDRV_OUT     binary dump: int32 {magic 0x52444c4c, n_prompt, n_gen, n_vocab}, int32 tokens[n_gen],
//                  float logits[n_gen][n_vocab] (logits that produced each generated token)
//   REFDRV_FORCE   comma-separated token ids fed instead of the argmax (teacher forcing; logits are still dumped)
//   REFDRV_SHIFT   "step,n_keep,n_discard": before generation step `step` the context is shifted the way examples/main/main.cpp does when it
//                  runs out of cells (llama_kv_cache_seq_rm + llama_kv_cache_seq_add, :583-:601): the next llama_decode runs the K-shift graph
//   REFDRV_RM      "step,p0,p1": before generation step `step` positions [p0, p1) of the sequence are removed (llama_kv_cache_seq_rm): holes
//   REFDRV_DEFRAG  "step": before generation step `step` llama_kv_cache_defrag + llama_kv_cache_update (build_defrag, src/llama.cpp:10721)
//   REFDRV_CHUNK   prompt tokens per llama_decode call (default: all = one prefill batch; clamped to n_batch)
//   REFDRV_LOUT    binary dump of every layer's output row of the FIRST single-token decode (the tensors llm_build_* names "l_out-<il>", src/llama.cpp:11177,
//                  and "result_norm"), captured through the scheduler's eval callback (cparams.cb_eval, ggml_backend_sched_set_eval_callback src/llama.cpp:18457):
//                  int32 {magic 0x54554f4c, n_rows, n_embd}, int32 layer[n_rows] (-1 = result_norm), float rows[n_rows][n_embd] - where two builds / backends
//                  part, layer by layer, before a flipped token can cascade
// Timing is wall-clock around llama_decode + llama_synchronize and the reference's llama_perf_context
// (src/llama.cpp:23832-23862), printed as one JSON line on stdout.
#include "arg.h"
#include "common.h"
#include "llama.h"
#include "ggml-backend.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static std::vector<int> parse_ids(const char * s) {
    std::vector<int> v;
    if (!s) return v;
    while (*s) {
        char * e = nullptr;
        long x = strtol(s, &e, 10);
        if (e == s) break;
        v.push_back((int) x);
        s = (*e == ',') ? e + 1 : e;
    }
    return v;
}

// per-layer capture (REFDRV_LOUT): armed around one llama_decode only
struct LoutCapture { bool armed = false; std::vector<int32_t> layer; std::vector<float> rows; int n_embd = 0; };
static bool lout_cb(struct ggml_tensor * t, bool ask, void * ud) {
    LoutCapture * c = (LoutCapture *) ud;
    if (!c->armed) return false;
    int il = -2;
    if (strncmp(t->name, "l_out-", 6) == 0) il = atoi(t->name + 6);
    else if (strcmp(t->name, "result_norm") == 0) il = -1;
    if (il == -2) return false;
    if (ask) return true;
    if (t->type != GGML_TYPE_F32) return true;
    const int64_t ne0 = t->ne[0], last = t->ne[1] - 1;              // the last token's row
    c->n_embd = (int) ne0;
    const size_t at = c->rows.size();
    c->rows.resize(at + (size_t) ne0);
    ggml_backend_tensor_get(t, c->rows.data() + at, (size_t) last * t->nb[1], (size_t) ne0 * sizeof(float));
    c->layer.push_back(il);
    return true;
}

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char ** argv) {
    gpt_params params;
    if (!gpt_params_parse(argc, argv, params, LLAMA_EXAMPLE_COMMON)) return 2;
    LoutCapture lout;
    if (getenv("REFDRV_LOUT")) { params.cb_eval = lout_cb; params.cb_eval_user_data = &lout; }
    llama_backend_init();
    llama_numa_init(params.numa);

    const double t_load0 = now_ms();
    llama_init_result init = llama_init_from_gpt_params(params);
    llama_model * model = init.model;
    llama_context * ctx = init.context;
    if (!model || !ctx) { fprintf(stderr, "refdrv: failed to initialise model/context\n"); return 3; }
    const double t_load = now_ms() - t_load0;

    std::vector<int> prompt = parse_ids(getenv("REFDRV_PROMPT"));
    if (prompt.empty()) prompt.push_back(1);
    const std::vector<int> force = parse_ids(getenv("REFDRV_FORCE"));
    const int n_gen = getenv("REFDRV_NGEN") ? atoi(getenv("REFDRV_NGEN")) : 16;
    const int n_vocab = llama_n_vocab(model);
    int chunk = getenv("REFDRV_CHUNK") ? atoi(getenv("REFDRV_CHUNK")) : (int) prompt.size();
    if (chunk < 1) chunk = 1;
    if (chunk > (int) llama_n_batch(ctx)) chunk = (int) llama_n_batch(ctx);      // llama_decode asserts n_tokens <= n_batch (and ggml_abort's gdb fork can hang)

    std::vector<llama_token> toks(prompt.begin(), prompt.end());
    std::vector<int32_t> gen;
    std::vector<float> all_logits;
    all_logits.reserve((size_t) n_gen * n_vocab);

    // ---- prompt
    const double t_p0 = now_ms();
    int n_past = 0;
    while (n_past < (int) toks.size()) {
        const int n = std::min(chunk, (int) toks.size() - n_past);
        if (llama_decode(ctx, llama_batch_get_one(toks.data() + n_past, n, n_past, 0))) { fprintf(stderr, "refdrv: llama_decode(prompt) failed\n"); return 4; }
        n_past += n;
    }
    llama_synchronize(ctx);
    const double t_prompt = now_ms() - t_p0;

    // ---- greedy generation (first maximum wins, like llama_sampler_greedy, src/llama-sampling.cpp:390-397)
    std::vector<double> step_ms;
    const std::vector<int> shift = parse_ids(getenv("REFDRV_SHIFT"));
    const std::vector<int> rm = parse_ids(getenv("REFDRV_RM")), defrag = parse_ids(getenv("REFDRV_DEFRAG"));
    for (int i = 0; i < n_gen; ++i) {
        const float * lg = llama_get_logits_ith(ctx, -1);
        int best = 0;
        for (int v = 1; v < n_vocab; ++v) if (lg[v] > lg[best]) best = v;
        gen.push_back(best);
        all_logits.insert(all_logits.end(), lg, lg + n_vocab);
        if (i == n_gen - 1) break;
        llama_token next = (i < (int) force.size()) ? force[i] : best;
        if (shift.size() == 3 && i == shift[0]) {
            const int n_keep = shift[1], n_discard = shift[2];
            llama_kv_cache_seq_rm (ctx, 0, n_keep, n_keep + n_discard);
            llama_kv_cache_seq_add(ctx, 0, n_keep + n_discard, n_past, -n_discard);
            n_past -= n_discard;
        }
        if (rm.size() == 3 && i == rm[0]) llama_kv_cache_seq_rm(ctx, 0, rm[1], rm[2]);
        if (defrag.size() == 1 && i == defrag[0]) { llama_kv_cache_defrag(ctx); llama_kv_cache_update(ctx); }
        const double t0 = now_ms();
        lout.armed = i == 0 && getenv("REFDRV_LOUT") != nullptr;
        if (llama_decode(ctx, llama_batch_get_one(&next, 1, n_past, 0))) { fprintf(stderr, "refdrv: llama_decode(step %d) failed\n", i); return 5; }
        llama_synchronize(ctx);
        lout.armed = false;
        step_ms.push_back(now_ms() - t0);
        n_past += 1;
    }

    if (const char * out = getenv("REFDRV_OUT")) {
        FILE * f = fopen(out, "wb");
        if (!f) { fprintf(stderr, "refdrv: cannot write %s\n", out); return 6; }
        const int32_t hdr[4] = {0x52444c4c, (int32_t) prompt.size(), (int32_t) gen.size(), n_vocab};
        fwrite(hdr, 4, 4, f);
        fwrite(gen.data(), 4, gen.size(), f);
        fwrite(all_logits.data(), 4, all_logits.size(), f);
        fclose(f);
    }

    if (const char * lo = getenv("REFDRV_LOUT")) {
        FILE * f = fopen(lo, "wb");
        if (!f) { fprintf(stderr, "refdrv: cannot write %s\n", lo); return 6; }
        const int32_t hdr[3] = {0x54554f4c, (int32_t) lout.layer.size(), lout.n_embd};
        fwrite(hdr, 4, 3, f);
        fwrite(lout.layer.data(), 4, lout.layer.size(), f);
        fwrite(lout.rows.data(), 4, lout.rows.size(), f);
        fclose(f);
    }

    // decode rate: drop the first min(5, n/2) steps like llama_perf (src/llama.cpp:3383 discards early evals)
    double sum = 0.0, best_ms = 1e30; int cnt = 0;
    const int skip = std::min<int>(5, (int) step_ms.size() / 2);
    for (size_t i = skip; i < step_ms.size(); ++i) { sum += step_ms[i]; best_ms = std::min(best_ms, step_ms[i]); ++cnt; }
    const llama_perf_context_data pd = llama_perf_context(ctx);
    printf("{\"refdrv\": 1, \"n_prompt\": %d, \"n_gen\": %d, \"n_vocab\": %d, \"ngl\": %d, \"threads\": %d, \"load_ms\": %.1f, "
           "\"prompt_ms\": %.3f, \"prompt_tok_s\": %.2f, \"decode_ms_avg\": %.4f, \"decode_ms_min\": %.4f, \"decode_tok_s\": %.3f, "
           "\"perf_t_eval_ms\": %.3f, \"perf_n_eval\": %d, \"perf_t_p_eval_ms\": %.3f, \"perf_n_p_eval\": %d}\n",
           (int) prompt.size(), (int) gen.size(), n_vocab, params.n_gpu_layers, params.cpuparams.n_threads, t_load,
           t_prompt, prompt.size() * 1000.0 / t_prompt, cnt ? sum / cnt : 0.0, cnt ? best_ms : 0.0, cnt ? 1000.0 * cnt / sum : 0.0,
           pd.t_eval_ms, pd.n_eval, pd.t_p_eval_ms, pd.n_p_eval);
    fflush(stdout);

    llama_free(ctx);
    llama_free_model(model);
    llama_backend_free();
    return 0;
}
