// Training-step closure around the hot path: Adam over the flat fp32 parameter /
// gradient buckets (one launch for all 640,967 parameters; reference: torch.optim.Adam(lr=2e-4) at
// gbm/classify_combined.py:519, stepped every few bags at :450-454) and a single-launch re-pack of every
// convolution filter into MFMA fragment order after the weights changed.
#include "pack.cuh"

// torch.optim.Adam semantics (no amsgrad, L2 weight decay folded into the gradient):
//   g += wd*p;  m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;
//   p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
__global__ void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                 float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps, float wd,
                                 float bc1, float bc2_sqrt, float grad_scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float gi = g[i] * grad_scale;
        const float pi = p[i];
        if (wd != 0.f) gi += wd * pi;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
    }
}

extern "C" int mil_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                             void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || step < 1) return MIL_ERR_ARG;
    if (n == 0) return MIL_OK;
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
    int grid = (int)((n + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(adam_step_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), params, grads,
                       exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, grad_scale);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

// ---------------------------------------------------------------------------------------------
// One launch packs every filter listed in a device-resident job table (index maps and job record: pack.cuh).
// blockIdx.y = job.
__global__ void pack_all_kernel(const PackJob* __restrict__ jobs) {
    const PackJob j = jobs[blockIdx.y];
    const int total = j.nsteps * j.NT * 64 * 8;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) mil_pack_job_elem(j, idx);
}

// Fills one host-side job record (the caller copies the table to the device once).
extern "C" int mil_pack_job_bytes(void) { return (int)sizeof(PackJob); }

extern "C" int mil_pack_job_fill(void* job_host, const float* w, const float* bias, void* out, float* bias_pad, int cout,
                                 int cin, int ks, int mode, int dtype) {
    if (!job_host || !w || !out || mode < 0 || mode > 3) return MIL_ERR_ARG;
    PackJob* j = reinterpret_cast<PackJob*>(job_host);
    *j = PackJob{};
    j->w = w; j->bias = bias; j->out = out; j->bias_pad = bias_pad;
    j->cout = cout; j->cin = cin; j->ks = ks; j->mode = mode; j->dtype = dtype;
    mil_pack_job_dims(j);
    return MIL_OK;
}

extern "C" int mil_pack_all(const void* jobs_device, int njobs, void* stream) {
    if (!jobs_device || njobs < 0) return MIL_ERR_ARG;
    if (njobs == 0) return MIL_OK;
    hipLaunchKernelGGL(pack_all_kernel, dim3(64, njobs), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const PackJob*>(jobs_device));
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}
