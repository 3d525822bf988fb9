// pi_infer_kernels.hip — batched inference-time interpolation on the solver's grids for gfx950.
//
// Device twin of the reference's CPU helper utils/barycentric.py (numba):
//   get_barycentric_weights_and_indices  :15-77    one thread per query point
//   get_optimal_action                   :80-112   weights @ action_space[policy[indices]]
// A device-code TEMPLATE: pi_infer.cpp prepends the handle's grid as compile-time constants
// (PI_D, PI_LO_INIT, PI_HI_INIT, PI_SHAPE_INIT, PI_STRIDES_INIT, PI_BITS_INIT = the caller's
// corner_bits, row-major) and hipRTC compiles it like the sweep kernels, with -ffp-contract=off.
// The arithmetic is the helper's, type for type (numba's typing of the body):
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
//
// Shape of the kernel (measured on MI355X, profiles/r03/inference.txt): the corner loop is fully
// unrolled over the constant table, so the 2^D policy look-ups of a query — random 4-byte reads of
// a table that does not fit any cache in 4-D / 6-D — are all in flight together and the action
// values follow as a second independent batch; weights and indices leave through an LDS transpose
// (XOR-swizzled against bank conflicts) so that a wave stores whole 256-byte runs instead of 64
// scattered words per instruction.

#define PI_INFER_C (1 << PI_D)
#define PI_INFER_BLOCK 256

struct PiInferGrid {
    float lo[PI_D], hi[PI_D];
    int shape[PI_D], stride[PI_D];
    int bits[PI_INFER_C][PI_D];
};
__device__ constexpr PiInferGrid PI_IG = {PI_LO_INIT, PI_HI_INIT, PI_SHAPE_INIT, PI_STRIDES_INIT, PI_BITS_INIT};

// rows of PI_INFER_C words, one per thread of the workgroup; word (row, col) lives at
// row * C + (col ^ (row % C)): a wave writing one column or reading one run of 64 consecutive
// words touches every bank at most twice
__device__ __forceinline__ unsigned int pi_infer_slot(unsigned int row, unsigned int col) {
    return row * PI_INFER_C + (col ^ (row & (PI_INFER_C - 1)));
}

// out[(k0 + row) * C + col] = rows[row][col] for the rows of this workgroup that exist (rows_here)
template <typename T>
__device__ __forceinline__ void pi_infer_store_rows(const T (&mine)[PI_INFER_C], unsigned int* lds,
                                                    T* __restrict__ out, long long k0, unsigned int rows_here) {
    const unsigned int tid = threadIdx.x;
    __syncthreads();                                   // the previous user of the LDS rows is done
#pragma unroll
    for (int c = 0; c < PI_INFER_C; ++c) lds[pi_infer_slot(tid, (unsigned int)c)] = __builtin_bit_cast(unsigned int, mine[c]);
    __syncthreads();
    const unsigned int words = rows_here * PI_INFER_C;
    unsigned int* dst = reinterpret_cast<unsigned int*>(out) + k0 * PI_INFER_C;
#pragma unroll
    for (int i = 0; i < PI_INFER_C; ++i) {
        const unsigned int j = (unsigned int)i * PI_INFER_BLOCK + tid;
        if (j < words) dst[j] = lds[pi_infer_slot(j / PI_INFER_C, j % PI_INFER_C)];
    }
}

extern "C" __global__ void __launch_bounds__(PI_INFER_BLOCK)
pi_infer_kernel(const float* __restrict__ pts, long long m, const int* __restrict__ policy,
                const float* __restrict__ actions, float* __restrict__ out_action, float* __restrict__ out_w,
                int* __restrict__ out_idx) {
    __shared__ unsigned int lds_rows[PI_INFER_BLOCK * PI_INFER_C];
    const long long k0 = (long long)blockIdx.x * PI_INFER_BLOCK;
    const unsigned int rows_here = (unsigned int)min((long long)PI_INFER_BLOCK, m - k0);
    const bool live = threadIdx.x < rows_here;
    const long long k = k0 + (live ? threadIdx.x : rows_here - 1);      // idle lanes shadow the last point
    int base[PI_D];
    double t[PI_D];
#pragma unroll
    for (int d = 0; d < PI_D; ++d) {
        const float l = PI_IG.lo[d], h = PI_IG.hi[d];
        const double step = (double)(h - l) / (double)(PI_IG.shape[d] - 1);
        const float x = pts[k * PI_D + d];
        const float p = fmaxf(l, fminf(x, h));
        const double cell = (double)(p - l) / step;
        int i = (int)cell;
        if (i >= PI_IG.shape[d] - 1) i = PI_IG.shape[d] - 2;
        base[d] = i;
        t[d] = (double)(float)(((double)p - ((double)l + (double)i * step)) / step);
    }
    float wf[PI_INFER_C];
    int flat[PI_INFER_C];
#pragma unroll
    for (int c = 0; c < PI_INFER_C; ++c) {
        double w = 1.0;
        int f = 0;
#pragma unroll
        for (int d = 0; d < PI_D; ++d) {
            w *= PI_IG.bits[c][d] ? t[d] : (1.0 - t[d]);
            f += (base[d] + PI_IG.bits[c][d]) * PI_IG.stride[d];
        }
        wf[c] = (float)w;
        flat[c] = f;
    }
    if (out_action != nullptr) {
        int a_idx[PI_INFER_C];
#pragma unroll
        for (int c = 0; c < PI_INFER_C; ++c) a_idx[c] = policy[flat[c]];
        float act = 0.0f;
#pragma unroll
        for (int c = 0; c < PI_INFER_C; ++c) {
            const float prod = wf[c] * actions[a_idx[c]];
            act = act + prod;
        }
        if (live) out_action[k] = act;
    }
    if (out_w != nullptr) pi_infer_store_rows(wf, lds_rows, out_w, k0, rows_here);
    if (out_idx != nullptr) pi_infer_store_rows(flat, lds_rows, out_idx, k0, rows_here);
}
