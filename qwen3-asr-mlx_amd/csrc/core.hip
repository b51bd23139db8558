// Library plumbing: error string, version, launch check, hipGraph capture helpers.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

extern "C" void q3asr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* q3asr_last_error(void) { return g_err; }
extern "C" int q3asr_version(void) { return 100; }          // 0.1.0
extern "C" const char* q3asr_arch(void) { return "gfx950"; }

// Diagnostic switches (tools/ab_step.py): every option selects between two launch forms with identical results; all default to 1.
int g_q3_option[8] = {1, 1, 1, 1, 1, 1, 1, 1};
extern "C" int q3asr_set_option(int option, int value) {
    if (option != Q3ASR_OPT_HEAD_XCD_MAP && option != Q3ASR_OPT_HEAD_STREAM && option != Q3ASR_OPT_ATTN_XCD_ORDER && option != Q3ASR_OPT_SAMPLER_RESIDENT && option != Q3ASR_OPT_ATTN_HALF_WG) { q3asr_set_error("q3asr_set_option: unknown option %d", option); return -1; }
    const int was = g_q3_option[option];
    g_q3_option[option] = value ? 1 : 0;
    return was;
}

int q3_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        q3asr_set_error("%s: %s", what, hipGetErrorString(e));
        return Q3_ELAUNCH;
    }
    return Q3_OK;
}

// ---- hipGraph capture of a decode step (the reference's per-token loop, generate.py:289-313,
// becomes one graph replay per token).  The caller launches kernels on `stream` between
// begin/end; nothing may allocate or synchronise in between.
extern "C" int q3asr_graph_begin(void* stream) {
    hipError_t e = hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { q3asr_set_error("graph_begin: %s", hipGetErrorString(e)); return Q3_ELAUNCH; }
    return Q3_OK;
}
extern "C" int q3asr_graph_end(void* stream, void** graph_exec_out) {
    Q3_REQUIRE(graph_exec_out, "graph_end: null output");
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture((hipStream_t)stream, &g);
    if (e != hipSuccess || !g) { q3asr_set_error("graph_end: %s", hipGetErrorString(e)); return Q3_ELAUNCH; }
    hipGraphExec_t ex = nullptr;
    e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) { q3asr_set_error("graph_instantiate: %s", hipGetErrorString(e)); return Q3_ELAUNCH; }
    *graph_exec_out = (void*)ex;
    return Q3_OK;
}
extern "C" int q3asr_graph_launch(void* graph_exec, void* stream) {
    Q3_REQUIRE(graph_exec, "graph_launch: null graph");
    hipError_t e = hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream);
    if (e != hipSuccess) { q3asr_set_error("graph_launch: %s", hipGetErrorString(e)); return Q3_ELAUNCH; }
    return Q3_OK;
}
extern "C" int q3asr_graph_destroy(void* graph_exec) {
    if (graph_exec) (void)hipGraphExecDestroy((hipGraphExec_t)graph_exec);
    return Q3_OK;
}

// ---- HIP-event timing on an explicit stream (bench.py's roofline leg: torch.cuda.Event only
// sees torch's current stream).
extern "C" int q3asr_event_create(void** ev) {
    Q3_REQUIRE(ev, "event_create: null output");
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) { q3asr_set_error("event_create failed"); return Q3_ELAUNCH; }
    *ev = (void*)e;
    return Q3_OK;
}
extern "C" int q3asr_event_record(void* ev, void* stream) {
    if (hipEventRecord((hipEvent_t)ev, (hipStream_t)stream) != hipSuccess) { q3asr_set_error("event_record failed"); return Q3_ELAUNCH; }
    return Q3_OK;
}
extern "C" int q3asr_event_elapsed_ms(void* start, void* stop, float* ms) {
    Q3_REQUIRE(ms, "event_elapsed: null output");
    if (hipEventSynchronize((hipEvent_t)stop) != hipSuccess || hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop) != hipSuccess) {
        q3asr_set_error("event_elapsed failed");
        return Q3_ELAUNCH;
    }
    return Q3_OK;
}
extern "C" int q3asr_event_destroy(void* ev) {
    if (ev) (void)hipEventDestroy((hipEvent_t)ev);
    return Q3_OK;
}
