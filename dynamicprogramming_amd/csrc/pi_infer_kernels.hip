// pi_infer_kernels.hip — batched inference-time interpolation on the solver's grids for gfx950.
//
// Device twin of the reference's CPU helper utils/barycentric.py (numba):
//   get_barycentric_weights_and_indices  :15-77    one thread per query point
//   get_optimal_action                   :80-112   weights @ action_space[policy[indices]]
// Compiled by hipRTC like the sweep kernels (pi_infer.cpp prepends `#define PI_D <D>`), with
// -ffp-contract=off.  The arithmetic is the helper's, type for type (numba's typing of the body):
//   step  = float64(hi - lo [float32 subtraction]) / (shape - 1)
//   p     = max(lo, min(point, hi))                            float32, the POINT is clamped
//   cell  = float64(p - lo [float32 subtraction]) / step;  idx = int(cell), capped at shape - 2
//   t     = float32((float64(p) - (float64(lo) + idx * step)) / step)
//   w[c]  = float32(1.0 * prod_d (bit ? float64(t_d) : 1.0 - float64(t_d)))   in dimension order
//   flat[c] = sum_d (idx_d + bit) * stride_d,   bit = corner_bits[c][d]  (caller's table: any order)
// IEEE float64 division, subtraction and multiplication are correctly rounded on the device as on
// the host, so indices and weights equal the helper's bit for bit (tests/test_gpu_endtoend.py against
// tests/golden/barycentric_utils.npz, vectors produced by the reference's own function).
// The action is the float32 sum over corners in ASCENDING corner order, multiply then add (numpy's
// `lambdas @ neighbor_actions` leaves the order to BLAS; the difference is at most a few ulp).

#define PI_INFER_C (1 << PI_D)

extern "C" __global__ void __launch_bounds__(256)
pi_infer_kernel(const float* __restrict__ pts, long long m, const float* __restrict__ lo,
                const float* __restrict__ hi, const int* __restrict__ shape, const int* __restrict__ strides,
                const int* __restrict__ bits, const int* __restrict__ policy, const float* __restrict__ actions,
                float* __restrict__ out_action, float* __restrict__ out_w, int* __restrict__ out_idx) {
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (k >= m) return;
    int base[PI_D];
    double t[PI_D];
#pragma unroll
    for (int d = 0; d < PI_D; ++d) {
        const float l = lo[d], h = hi[d];
        const double step = (double)(h - l) / (double)(shape[d] - 1);
        const float x = pts[k * PI_D + d];
        const float p = fmaxf(l, fminf(x, h));
        const double cell = (double)(p - l) / step;
        int i = (int)cell;
        if (i >= shape[d] - 1) i = shape[d] - 2;
        base[d] = i;
        t[d] = (double)(float)(((double)p - ((double)l + (double)i * step)) / step);
    }
    float act = 0.0f;
    for (int c = 0; c < PI_INFER_C; ++c) {
        double w = 1.0;
        int flat = 0;
#pragma unroll
        for (int d = 0; d < PI_D; ++d) {
            const int bit = bits[c * PI_D + d];
            w *= bit ? t[d] : (1.0 - t[d]);
            flat += (base[d] + bit) * strides[d];
        }
        const float wf = (float)w;
        if (out_w) out_w[k * PI_INFER_C + c] = wf;
        if (out_idx) out_idx[k * PI_INFER_C + c] = flat;
        if (out_action) {
            const float prod = wf * actions[policy[flat]];
            act = act + prod;
        }
    }
    if (out_action) out_action[k] = act;
}
