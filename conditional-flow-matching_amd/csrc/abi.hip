// abi.hip — version + workspace sizing for libcfm_gfx950.so
#include "cfm_common.h"

extern "C" size_t cfm_sk_ws_bytes_internal(int B0, int B1);
extern "C" size_t cfm_asg_ws_bytes_internal(int n);
extern "C" size_t cfm_assign_batch_ws_bytes_internal(int n, int nb);
extern "C" size_t cfm_sd_ws_bytes_internal(int B0, int B1);
extern "C" size_t cfm_mlp_ws_bytes_internal(int B, int width);
extern "C" size_t cfm_ode_ws_bytes_internal(int B, int width, int d);
extern "C" size_t cfm_ub_ws_bytes_internal(int B0, int B1);
extern "C" size_t cfm_cost_ws_bytes_internal(int B0, int B1, int d);
extern "C" size_t cfm_mlp_train_ws_bytes_internal(int B, int maxw, int max_params);
extern "C" size_t cfm_tp_ws_bytes_internal(int B0, int B1);

extern "C" int cfm_abi_version(void) { return CFM_ABI_VERSION; }

extern "C" size_t cfm_workspace_bytes(int op, int B0, int B1, int d) {
    if (B0 < 0 || B1 < 0) return 0;
    switch (op) {
        case CFM_OP_SINKHORN: return cfm_align_up(cfm_sk_ws_bytes_internal(B0, B1), 256);
        case CFM_OP_ASSIGN:          // d > 1: a batch of d problems (cfm_assign_exact_batch_f32)
            return d > 1 ? cfm_align_up(cfm_assign_batch_ws_bytes_internal(B0 > B1 ? B0 : B1, d), 256)
                         : cfm_align_up(cfm_asg_ws_bytes_internal(B0 > B1 ? B0 : B1), 256);
        case CFM_OP_SAMPLE_DENSE: return cfm_align_up(cfm_sd_ws_bytes_internal(B0, B1), 256);
        case CFM_OP_MLP: return cfm_align_up(cfm_mlp_ws_bytes_internal(B0, B1), 256);
        case CFM_OP_ODE: return cfm_align_up(cfm_ode_ws_bytes_internal(B0, B1, d), 256);
        case CFM_OP_UNBALANCED: return cfm_align_up(cfm_ub_ws_bytes_internal(B0, B1), 256);
        case CFM_OP_COST: return d > 0 ? cfm_align_up(cfm_cost_ws_bytes_internal(B0, B1, d), 256) : 0;
        case CFM_OP_MLP_TRAIN: return cfm_align_up(cfm_mlp_train_ws_bytes_internal(B0, B1, d), 256);
        case CFM_OP_TRANSPORT: return (B0 > 0 && B1 > 0) ? cfm_align_up(cfm_tp_ws_bytes_internal(B0, B1), 256) : 0;
        default: return 0;
    }
}

// ---- runtime: CU-partitioned streams (hipExtStreamCreateWithCUMask) ----
extern "C" int cfm_stream_create_cu_mask(const uint32_t* cu_mask, int n_words, void** stream) {
    if (!cu_mask || !stream || n_words < 1 || n_words > 32) return CFM_EINVAL;
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return CFM_EINVAL;
    const int ncu = prop.multiProcessorCount, nx = 8;
    int per_xcd[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < ncu && i < 32 * n_words; ++i)
        if (cu_mask[i >> 5] >> (i & 31) & 1u) per_xcd[i % nx]++;
    for (int x = 0; x < nx && x < ncu; ++x)
        if (per_xcd[x] == 0) return CFM_EINVAL;           // an XCD without CUs: its share of every grid would never run
    hipStream_t s = nullptr;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, cu_mask);
    if (e != hipSuccess) return (int)e;
    *stream = (void*)s;
    return 0;
}
extern "C" int cfm_stream_destroy(void* stream) {
    if (!stream) return CFM_EINVAL;
    return cfm_hip(hipStreamDestroy((hipStream_t)stream));
}
