// Error reporting, version and device discovery for libgsn_hip.so (C ABI: gsn_last_error, gsn_version, gsn_device_count).
#include <hip/hip_runtime.h>

#include <cstring>

#include "gsn_internal.h"

namespace gsn {

static thread_local char g_err[512] = "";

int set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace gsn

extern "C" const char *gsn_last_error(void) { return gsn::g_err; }

extern "C" int gsn_version(void) { return GSN_ABI_VERSION; }

extern "C" int gsn_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    int good = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && std::strncmp(p.gcnArchName, "gfx950", 6) == 0) ++good;
    }
    return good;
}

// 0 when `stream` is not being captured into a graph, else the capture's id (unique per capture sequence, hipStreamGetCaptureInfo).
// The host side keys its zero-initialised scratch arenas on it: a buffer whose fill was recorded into one capture must not be
// handed out in another capture or in eager execution (the fill would not run there).
extern "C" int64_t gsn_stream_capture_id(void *stream) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    if (hipStreamGetCaptureInfo(reinterpret_cast<hipStream_t>(stream), &st, &id) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    if (st == hipStreamCaptureStatusNone) return 0;
    return (int64_t)(id ? id : 1);
}

namespace gsn {
int current_device() {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    return dev;
}
}  // namespace gsn

// HP-1 + HP-2 in one host call (include/gsn_abi.h: gsn_count_layer_step_hip): the counting launch with its side outputs, then layer 0 on the packs
extern "C" int gsn_count_layer_step_hip(const gsn_count_call *c, const gsn_layer_pack16_call *l, void *event_between, void *stream) {
    if (!c || !l) return gsn::set_error(GSN_E_INVALID, "gsn_count_layer_step_hip: null call struct");
    const int rc = gsn_count_encode_pack16_side_hip(c->plan_host, c->plan_dev, c->plan_words, c->n_graphs, c->node_ptr, c->edge_ptr, c->edge_index,
                                                    c->edge_row_stride, c->ids_are_global, c->max_nodes, c->max_edges, c->out, c->status, c->n_classes,
                                                    c->clamp, c->pack, c->pack_stride, c->pack_col0, c->side, stream);
    if (rc != GSN_OK) return rc;
    if (event_between && hipEventRecord(reinterpret_cast<hipEvent_t>(event_between), reinterpret_cast<hipStream_t>(stream)) != hipSuccess)
        return gsn::set_error(GSN_E_HIP, "gsn_count_layer_step_hip: hipEventRecord(event_between)");
    return gsn_layer_fused_fwd_pack16_hip(l->n_nodes, l->n_edges, l->seg_ptr, l->edge, l->x, l->d_x, l->node0, l->node1, l->prepared, l->pack,
                                          l->edge_rows, l->out, stream);
}
