// msm_tables.hip.h — the window tables of the fixed-base mode, table[w][i] = 2^(c w) P_i, and the identity test of a basis
// (part of msm.hip's translation unit, inside namespace zk)
// ------------------------------------------------------ fixed-base tables ---
// table[w][i] = 2^(c w) * P_i (affine).  One launch per window: c doublings + one inversion.
__global__ __launch_bounds__(64) void msm_table_step_kernel(const G1Affine* __restrict__ prev, G1Affine* __restrict__ next,
                                                            uint32_t n, uint32_t c) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const G1Affine p = affine_load(prev + i);
    G1Affine r;
    if (affine_is_identity(p)) {
        r.x = Fq::zero();
        r.y = Fq::zero();
    } else {
        G1X acc = g1x_dbl_affine(p.x, p.y);
        for (uint32_t k = 1; k < c; k++) acc = g1x_dbl(acc);
        if (acc.is_identity()) {  // cannot happen on a prime-order curve; kept for completeness
            r.x = Fq::zero();
            r.y = Fq::zero();
        } else {
            const Fq t = fe_inv(acc.zzz);
            const Fq u = fe_mul(acc.zz, t);
            r.x = fe_mul(acc.x, fe_sqr(u));
            r.y = fe_mul(acc.y, t);
        }
    }
    fe_store(&next[i].x, r.x);
    fe_store(&next[i].y, r.y);
}

// does any of the n points equal the identity (0, 0)?  Decides which accumulation loop a basis gets (msm_accumulate_kernel).
__global__ void msm_identity_flag_kernel(const G1Affine* __restrict__ b, uint32_t n, uint32_t* __restrict__ flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && affine_is_identity(affine_load(b + i))) atomicOr(flag, 1u);
}
hipError_t msm_bases_have_identity(const G1Affine* bases, uint32_t n, hipStream_t st, uint32_t* d_word, uint32_t* h_word, bool* out) {
    hipError_t e = hipMemsetAsync(d_word, 0, 4, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(msm_identity_flag_kernel, dim3((n + 255) / 256), dim3(256), 0, st, bases, n, d_word);
    if ((e = hipMemcpyAsync(h_word, d_word, 4, hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
    *out = *h_word != 0;
    return hipSuccess;
}

// x * 2^256 (standard memory form) -> x * 2^261 (the accumulation's internal form, canonical words): times 32
__global__ void msm_table_internal_kernel(G1Affine* __restrict__ t, size_t count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    G1Affine p = affine_load(t + i);
    for (int k = 0; k < 5; k++) {
        p.x = fe_add(p.x, p.x);
        p.y = fe_add(p.y, p.y);
    }
    fe_store(&t[i].x, p.x);
    fe_store(&t[i].y, p.y);
}

// the same rule msm_run applies (wide workspace && table stride == workspace length; workspaces are >= 1024 long)
bool msm_table_is_internal(uint32_t c, size_t n) { return msm_wide_applies(c, n) && n >= 1024; }

hipError_t msm_build_table(const G1Affine* bases, uint32_t n, uint32_t c, G1Affine* table, hipStream_t st) {
    const uint32_t nwin = nwin_for(c);
    hipError_t e = hipMemcpyAsync(table, bases, (size_t)n * sizeof(G1Affine), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return e;
    for (uint32_t w = 1; w < nwin; w++)
        hipLaunchKernelGGL(msm_table_step_kernel, dim3((n + 63) / 64), dim3(64), 0, st, table + (size_t)(w - 1) * n,
                           table + (size_t)w * n, n, c);
    if (msm_table_is_internal(c, n)) {
        // the wide path reads its window tables in the accumulation's internal form (the identity stays (0, 0))
        const size_t count = (size_t)nwin * n;
        hipLaunchKernelGGL(msm_table_internal_kernel, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, st, table, count);
    }
    return hipGetLastError();
}

