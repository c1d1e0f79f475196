// A whole GMRES cycle of the scalar annular solve in ONE launch (included by annular.hip, inside its
// anonymous namespace, after the vector kernels and fft_pair_kernel).
//
// The host-driven cycle (gmres_solve) is ten dependent launches per inner iteration — preconditioner,
// three radial mixes around two transform pairs, CGS2 in four — of 5-12 us each with 2-5 us between
// them: ~100 us per iteration for ~15 us of work (profiles/r02_poisson_2048_solve_budget.json), and
// one look at the Hessenberg column by the host.  Here a few dozen workgroups stay resident for the
// cycle; the stages of an iteration are separated by grid-wide barriers, the Arnoldi / Givens
// bookkeeping runs on the device (every workgroup repeats the few scalar operations on the same
// inputs, so all agree on when to stop without another exchange), and the host sees one result record.
//
// Work split.  Column stages: thread = (column c of the (M, n) unknown, one of RS row groups), a
// workgroup owns T / RS adjacent columns.  Everything per-column — preconditioner blocks, the radial
// matrices, w, the Krylov columns V_i[:, c] — is touched by that one workgroup only, for the whole
// cycle: no exchange, w lives in registers, z in LDS.  Row stages (the transform pairs): workgroup b
// takes row b.  Only the rows handed between the two kinds of stage (T, U: 2.5 + 1.2 MB per
// iteration at n = 4096) and the partial sums of the inner products cross workgroups.
//
// Hand-over (CDNA4 guide, Guideline 16): every handed-over value is written with 8-byte agent-scope
// relaxed atomic stores (write-through), every wave drains its stores (s_waitcnt vmcnt(0)) before the
// workgroup's lane 0 adds to the barrier counter, the counter is polled relaxed, and EVERY load of a
// handed-over value is an agent-scope relaxed atomic load (no L1 copy): no fence, no dependence on
// placement or dispatch order.  Spins are bounded; a time-out sets a sticky word, every workgroup
// leaves, and the host falls back to the launch-per-stage cycle.  The grid is at most 2 (M - 1) or
// n RS / T workgroups (64 at n = 4096) of <= 110 KB LDS: resident on any MI355X that is not
// oversubscribed by other persistent kernels; the counter is zeroed by a memset node per launch.
//
// Arithmetic: operator and preconditioner are the launch-per-stage kernels' sums in their order
// (bitwise the same operator); the inner products are summed per workgroup and then over the
// workgroups in index order — deterministic, but a different order from multidot_kernel's, so the
// Hessenberg entries differ from the host-driven cycle's in the last bits.
#pragma once

constexpr int PG_RS = 4;          // row groups per column
constexpr int PG_MR_MAX = 8;      // rows of a column per thread at most: M <= 32
constexpr int PG_RMAX = 32;       // longest cycle kept on the device
constexpr unsigned PG_SPIN_LIMIT = 1u << 24;

struct PgArgs {
    int M, restart, maxiter, G;
    double tol;
    const double *R01, *R12, *D01, *D12, *Bmat, *Kt;
    const cd* iks;
    const double *psi1, *ipsi1, *ipsi2;
    const fftcore::cd* tw;
    cd *T, *U;          // handed-over rows: (2 (M - 1), n), (M - 2, n)
    cd* V;              // Krylov basis (restart + 1, M n)
    const cd* b;
    cd* x;
    cd* part;           // (G, PG_RMAX + 2) partial inner products
    double* partn;      // (G) partial norms
    unsigned* counter;  // barrier counter (zeroed per launch), [1]: sticky time-out word
    double* result;     // [iterations, relative residual, converged (1) / cycle exhausted (0), bnorm]
    // LDS carve (in cd units from the base)
    int off_colA, off_red, off_hs, off_H, off_g, off_mat, lds_cd;
};

struct PgSync {
    ann_gu32* counter;
    ann_gu32* abort_word;
    unsigned G, epoch;
};

// grid-wide barrier; false: timed out somewhere (every workgroup then leaves the kernel)
__device__ __forceinline__ bool pg_grid_sync(PgSync& s, volatile int* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every wave: its handed-over stores have left
    __syncthreads();
    s.epoch += 1;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(s.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = s.epoch * s.G;
        int ok = 1;
        for (unsigned spins = 0;;) {
            if (__hip_atomic_load(s.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023u) == 0) {
                if (__hip_atomic_load(s.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                    ok = 0;
                    break;
                }
                if (spins >= PG_SPIN_LIMIT) {
                    __hip_atomic_store(s.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = 0;
                    break;
                }
            }
        }
        *flag = ok;
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // (no instruction: keeps later loads below the poll)
    return *flag != 0;
}

__device__ __forceinline__ void st_cd_agent(cd* p, cd v) {
    st_agent((double*)p, v.x);
    st_agent((double*)p + 1, v.y);
}
__device__ __forceinline__ cd ld_cd_agent(const cd* p) {
    return cd{ld_agent((const double*)p), ld_agent((const double*)p + 1)};
}

// rows -> inverse FFT -> field / n -> forward FFT of ONE handed-over row (fft_pair_kernel's arithmetic)
// (not inlined: the transform's ~200 registers stay out of the cycle kernel's own allocation)
template <int N>
__device__ __attribute__((noinline)) void pg_fft_pair_row(cd* row, const double* __restrict__ F, double s,
                                                const fftcore::cd* __restrict__ tw, fftcore::cd* buf, int t) {
    using G = fftcore::Cfg<N>;
    constexpr int T = G::T, P = G::P;
    const fftcore::PassTw<N> pw = fftcore::load_pass_twiddles<N>(t, tw);
    fftcore::cd v[P];
#pragma unroll
    for (int q = 0; q < P; ++q) {
        const cd a = ld_cd_agent(row + t + T * q);
        v[q] = fftcore::cd{a.x, a.y};
    }
    fftcore::fft_regs<N, +1, (T == 64)>(v, t, pw, buf);
    fftcore::lds_sync<(T == 64)>();
#pragma unroll
    for (int q = 0; q < P; ++q) {
        const double f = s * F[t + T * q];
        v[q] = fftcore::cd{v[q].x * f, v[q].y * f};
    }
    fftcore::fft_regs<N, -1, (T == 64)>(v, t, pw, buf);
#pragma unroll
    for (int q = 0; q < P; ++q) st_cd_agent(row + t + T * q, cd{v[q].x, v[q].y});
}

// (one workgroup per CU by its LDS: a wave may take the whole 512-entry register file of its SIMD)
template <int N, int MR>
__global__ __launch_bounds__(fftcore::Cfg<N>::T) __attribute__((amdgpu_waves_per_eu(1, 1)))
void gmres_scalar_persistent(PgArgs A) {
    constexpr int PG_MR = MR;         // rows of a column per thread: M <= PG_RS * MR
    using FG = fftcore::Cfg<N>;
    constexpr int T = FG::T, RS = PG_RS, CPB = T / RS, NW = (T + 63) / 64;
    constexpr int n = N;
    extern __shared__ double2 pg_lds[];
    fftcore::cd* fbuf = (fftcore::cd*)pg_lds;     // the transform pairs' exchange buffer ...
    cd* colB = (cd*)pg_lds;                       // ... and, between them, the T columns of this workgroup
    cd* colA = (cd*)pg_lds + A.off_colA;          // v / z / t columns: (M, CPB)
    cd* red = (cd*)pg_lds + A.off_red;            // (NW, PG_RMAX + 2) wave partials
    cd* hs = (cd*)pg_lds + A.off_hs;              // h1 (PG_RMAX + 2), h2 (PG_RMAX + 2)
    cd* Hm = (cd*)pg_lds + A.off_H;               // Hessenberg columns (PG_RMAX + 1) x PG_RMAX, thread 0 only
    cd* gcs = (cd*)pg_lds + A.off_g;              // Givens cosines, sines, right-hand side, y
    cd* gsn = gcs + (PG_RMAX + 1);
    cd* ggv = gsn + (PG_RMAX + 1);
    cd* gy = ggv + (PG_RMAX + 1);
    double* sc = (double*)(gy + (PG_RMAX + 1));   // [0] norm^2 / hn, [1] residual, [2] stop flag
    volatile int* bflag = (volatile int*)(sc + 6);
    // the radial matrices, once per launch: R01, D01 (M - 1, M), R12, D12 (M - 2, M - 1), B (M, M)
    double* mR01 = (double*)((cd*)pg_lds + A.off_mat);
    const int tid = threadIdx.x, blk = blockIdx.x, lane = tid & 63, wv = tid >> 6;
    const int M = A.M, m1 = M - 1, m2 = M - 2;
    const int64_t NB = (int64_t)M * n;
    const int cl = tid % CPB, part = tid / CPB;
    const int c = blk * CPB + cl;
    const bool has_col = c < n;
    const int R = A.restart < A.maxiter ? A.restart : A.maxiter;
    constexpr int PS = PG_RMAX + 2;
    double* mD01 = mR01 + m1 * M;
    double* mR12 = mD01 + m1 * M;
    double* mD12 = mR12 + m2 * m1;
    double* mB = mD12 + m2 * m1;
    for (int i = tid; i < m1 * M; i += T) {
        mR01[i] = A.R01[i];
        mD01[i] = A.D01[i];
    }
    for (int i = tid; i < m2 * m1; i += T) {
        mR12[i] = A.R12[i];
        mD12[i] = A.D12[i];
    }
    for (int i = tid; i < M * M; i += T) mB[i] = A.Bmat[i];
    __syncthreads();
    PgSync sy{(ann_gu32*)A.counter, (ann_gu32*)(A.counter + 1), (unsigned)A.G, 0u};

    auto wave_sum = [&](double v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        return v;
    };
    // sum over the workgroup of one double per thread, to every thread (waves in index order)
    auto block_sum = [&](double v) {
        v = wave_sum(v);
        __syncthreads();
        if (lane == 0) red[wv].x = v;
        __syncthreads();
        double s = red[0].x;
        for (int w = 1; w < NW; ++w) s += red[w].x;
        return s;
    };
    // inner products <V_i, w> over this workgroup's columns, i <= j, into part[blk][i]
    cd wcol[PG_MR];
    auto dots_to_partials = [&](int j) {
        for (int i0 = 0; i0 <= j; i0 += 4) {
            cd va[4][PG_MR];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int q = 0; q < PG_MR; ++q) {
                    const int r = part + RS * q;
                    va[u][q] = (has_col && i0 + u <= j && r < M) ? A.V[(size_t)(i0 + u) * NB + (size_t)r * n + c]
                                                                 : cd{0.0, 0.0};
                }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i0 + u > j) break;
                double sr = 0.0, si = 0.0;
#pragma unroll
                for (int q = 0; q < PG_MR; ++q) {
                    const int r = part + RS * q;
                    if (has_col && r < M) {
                        const cd a = va[u][q], b = wcol[q];
                        sr = fma(a.x, b.x, sr);
                        sr = fma(a.y, b.y, sr);
                        si = fma(a.x, b.y, si);
                        si = fma(-a.y, b.x, si);
                    }
                }
                sr = wave_sum(sr);
                si = wave_sum(si);
                if (lane == 0) red[wv * PS + i0 + u] = cd{sr, si};
            }
        }
        __syncthreads();
        if (tid <= j) {
            cd s = red[tid];
            for (int w = 1; w < NW; ++w) {
                s.x += red[w * PS + tid].x;
                s.y += red[w * PS + tid].y;
            }
            st_cd_agent(A.part + (size_t)blk * PS + tid, s);
        }
    };
    // h_i = sum over the workgroups of the partials -> dst[i], i <= j: wave w takes i = w, w + NW, ...,
    // lane g the partial of workgroup g (G <= 64), a fixed shuffle tree adds them
    auto gather_h = [&](int j, cd* dst) {
        for (int i = wv; i <= j; i += NW) {
            cd p{0.0, 0.0};
            if (lane < A.G) p = ld_cd_agent(A.part + (size_t)lane * PS + i);
            const double sr = wave_sum(p.x), si = wave_sum(p.y);
            if (lane == 0) dst[i] = cd{sr, si};
        }
        __syncthreads();
    };
    // w -= sum_i h_i V_i
    auto subtract = [&](int j, const cd* h) {
        if (!has_col) return;
        for (int i0 = 0; i0 <= j; i0 += 4) {
            cd va[4][PG_MR];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int q = 0; q < PG_MR; ++q) {
                    const int r = part + RS * q;
                    va[u][q] = (i0 + u <= j && r < M) ? A.V[(size_t)(i0 + u) * NB + (size_t)r * n + c] : cd{0.0, 0.0};
                }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i0 + u > j) break;
                const cd ci = h[i0 + u];
#pragma unroll
                for (int q = 0; q < PG_MR; ++q) {
                    const int r = part + RS * q;
                    if (r < M) {
                        const cd v = va[u][q];
                        wcol[q].x -= ci.x * v.x - ci.y * v.y;
                        wcol[q].y -= ci.x * v.y + ci.y * v.x;
                    }
                }
            }
        }
    };
    // colA <- the column whose rows this thread holds in `rows`; then out rows = Kinv column
    auto precondition = [&](const cd (&rows)[PG_MR], cd (&out)[PG_MR]) {
        __syncthreads();
        if (has_col) {
#pragma unroll
            for (int q = 0; q < PG_MR; ++q) {
                const int r = part + RS * q;
                if (r < M) colA[r * CPB + cl] = rows[q];
            }
        }
        __syncthreads();
        if (has_col) {
#pragma unroll
            for (int q = 0; q < PG_MR; ++q) {      // (row by row: one row's table entries in registers at a time)
                const int r = part + RS * q;
                __builtin_amdgcn_sched_barrier(0);
                if (r < M) {
                    // (one wave per SIMD: the row's M table entries are requested together, then summed
                    // in index order as prec_scalar_kernel does)
                    const double* K = A.Kt + (size_t)r * M * n + c;
                    double kw[PG_RS * PG_MR];
#pragma unroll
                    for (int k = 0; k < PG_RS * PG_MR; ++k) kw[k] = k < M ? K[(size_t)k * n] : 0.0;
                    double sr = 0.0, si = 0.0;
#pragma unroll
                    for (int k = 0; k < PG_RS * PG_MR; ++k) {
                        if (k < M) {
                            const cd v = colA[k * CPB + cl];
                            sr = fma(kw[k], v.x, sr);
                            si = fma(kw[k], v.y, si);
                        }
                    }
                    out[q] = cd{sr, si};
                }
            }
        }
        __syncthreads();
    };

    // ---- ||b||, v_0 = b / ||b|| ---------------------------------------------------------------
    cd vv[PG_MR];
    double s2 = 0.0;
#pragma unroll
    for (int q = 0; q < PG_MR; ++q) {
        const int r = part + RS * q;
        vv[q] = cd{0.0, 0.0};
        if (has_col && r < M) {
            vv[q] = A.b[(size_t)r * n + c];
            s2 = fma(vv[q].x, vv[q].x, s2);
            s2 = fma(vv[q].y, vv[q].y, s2);
        }
    }
    {
        const double bs = block_sum(s2);
        if (tid == 0) st_agent(A.partn + blk, bs);
    }
    if (!pg_grid_sync(sy, bflag)) return;
    if (wv == 0) {
        const double t = wave_sum(lane < A.G ? ld_agent(A.partn + lane) : 0.0);
        if (lane == 0) sc[0] = t;
    }
    __syncthreads();
    const double bnorm = sqrt(sc[0]);
    if (!(bnorm > 0.0)) {        // (uniform over the grid) x = 0
        if (has_col)
#pragma unroll
            for (int q = 0; q < PG_MR; ++q) {
                const int r = part + RS * q;
                if (r < M) A.x[(size_t)r * n + c] = cd{0.0, 0.0};
            }
        if (blk == 0 && tid == 0) {
            A.result[0] = 0.0;
            A.result[1] = 0.0;
            A.result[2] = 1.0;
            A.result[3] = 0.0;
        }
        return;
    }
    {
        const double s = 1.0 / bnorm;
#pragma unroll
        for (int q = 0; q < PG_MR; ++q) vv[q] = cd{s * vv[q].x, s * vv[q].y};
    }
    if (tid == 0) {
        ggv[0] = cd{bnorm, 0.0};
        for (int i = 1; i <= PG_RMAX; ++i) ggv[i] = cd{0.0, 0.0};
    }
    int iters = 0;
    double resid = 1.0;
    bool converged = false;
    // (workgroup 0 keeps a stage profile of the cycle in result[8 ..]: 100 MHz ticks per stage kind)
    unsigned long long stamp_t = wall_clock64();
    double stage_ticks[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto stamp = [&](int k) {
        const unsigned long long now = wall_clock64();
        stage_ticks[k] += (double)(now - stamp_t);
        stamp_t = now;
    };
    for (int j = 0; j < R; ++j) {
        // ---- column stage 0: V_j, z = Kinv v_j, T1 = R01 (z iks), T2 = D01 z --------------------
        if (has_col) {
#pragma unroll
            for (int q = 0; q < PG_MR; ++q) {
                const int r = part + RS * q;
                if (r < M) A.V[(size_t)j * NB + (size_t)r * n + c] = vv[q];
            }
        }
        cd zr[PG_MR];
        precondition(vv, zr);
        if (has_col) {
#pragma unroll
            for (int q = 0; q < PG_MR; ++q) {
                const int r = part + RS * q;
                if (r < M) colA[r * CPB + cl] = zr[q];      // the z column stays here until column stage 2
            }
        }
        __syncthreads();
        if (has_col) {
            const cd ik = A.iks[c];
            for (int ro = part; ro < m1; ro += RS) {
                const double* a1 = mR01 + ro * M;
                const double* a2 = mD01 + ro * M;
                double s1r = 0.0, s1i = 0.0, s2r = 0.0, s2i = 0.0;
#pragma unroll 4
                for (int k = 0; k < M; ++k) {
                    const cd v = colA[k * CPB + cl];
                    s1r = fma(a1[k], v.x, s1r);
                    s1i = fma(a1[k], v.y, s1i);
                    s2r = fma(a2[k], v.x, s2r);
                    s2i = fma(a2[k], v.y, s2i);
                }
                st_cd_agent(A.T + (size_t)ro * n + c, cmul(cd{s1r, s1i}, ik));
                st_cd_agent(A.T + (size_t)(m1 + ro) * n + c, cd{s2r, s2i});
            }
        }
        stamp(0);
        if (!pg_grid_sync(sy, bflag)) return;
        stamp(7);
        // ---- row stage 1: transform pairs of the 2 (M - 1) rows of T -----------------------------
        if (blk < 2 * m1)
            pg_fft_pair_row<N>(A.T + (size_t)blk * n,
                               blk < m1 ? A.ipsi1 + (size_t)blk * n : A.psi1 + (size_t)(blk - m1) * n, 1.0 / n,
                               A.tw, fbuf, tid);
        stamp(1);
        if (!pg_grid_sync(sy, bflag)) return;
        stamp(7);
        // ---- column stage 1: S = R12 (T1 iks) + D12 T2 --------------------------------------------
        if (has_col) {
            // (up to 2 PG_MR x 2 handed-over rows per thread, requested together)
            cd tv[2 * PG_MR];
#pragma unroll
            for (int u = 0; u < 2 * PG_MR; ++u) {
                const int k = part + RS * u;
                tv[u] = k < 2 * m1 ? ld_cd_agent(A.T + (size_t)k * n + c) : cd{0.0, 0.0};
            }
#pragma unroll
            for (int u = 0; u < 2 * PG_MR; ++u) {
                const int k = part + RS * u;
                if (k < 2 * m1) colB[k * CPB + cl] = tv[u];
            }
        }
        __syncthreads();
        if (has_col) {
            const cd ik = A.iks[c];
            for (int ro = part; ro < m2; ro += RS) {
                const double* a1 = mR12 + ro * m1;
                const double* a2 = mD12 + ro * m1;
                double s1r = 0.0, s1i = 0.0, s2r = 0.0, s2i = 0.0;
#pragma unroll 4
                for (int k = 0; k < m1; ++k) {
                    const cd v1 = colB[k * CPB + cl], v2 = colB[(m1 + k) * CPB + cl];
                    s1r = fma(a1[k], v1.x, s1r);
                    s1i = fma(a1[k], v1.y, s1i);
                    s2r = fma(a2[k], v2.x, s2r);
                    s2i = fma(a2[k], v2.y, s2i);
                }
                const cd t0 = cmul(cd{s1r, s1i}, ik);
                st_cd_agent(A.U + (size_t)ro * n + c, cd{fma(1.0, t0.x, s2r), fma(1.0, t0.y, s2i)});
            }
        }
        stamp(2);
        if (!pg_grid_sync(sy, bflag)) return;
        stamp(7);
        // ---- row stage 2: transform pairs of the M - 2 rows of U ----------------------------------
        if (blk < m2) pg_fft_pair_row<N>(A.U + (size_t)blk * n, A.ipsi2 + (size_t)blk * n, 1.0 / n, A.tw, fbuf, tid);
        stamp(3);
        if (!pg_grid_sync(sy, bflag)) return;
        stamp(7);
        // ---- column stage 2: w = B z - [luh]; first Gram-Schmidt pass ------------------------------
        if (has_col) {
#pragma unroll
            for (int q = 0; q < PG_MR; ++q) {
                const int r = part + RS * q;
                if (r < M) {
                    const double* a = mB + r * M;
                    double sr = 0.0, si = 0.0;
#pragma unroll 4
                    for (int k = 0; k < M; ++k) {
                        const cd v = colA[k * CPB + cl];
                        sr = fma(a[k], v.x, sr);
                        si = fma(a[k], v.y, si);
                    }
                    if (r < m2) {
                        const cd u = ld_cd_agent(A.U + (size_t)r * n + c);
                        sr = fma(-1.0, u.x, sr);
                        si = fma(-1.0, u.y, si);
                    }
                    wcol[q] = cd{sr, si};
                }
            }
        }
        __syncthreads();
        dots_to_partials(j);
        stamp(4);
        if (!pg_grid_sync(sy, bflag)) return;
        stamp(7);
        gather_h(j, hs);
        subtract(j, hs);
        // ---- second pass --------------------------------------------------------------------------
        dots_to_partials(j);
        stamp(5);
        if (!pg_grid_sync(sy, bflag)) return;
        stamp(7);
        gather_h(j, hs + PS);
        subtract(j, hs + PS);
        {
            double s = 0.0;
            if (has_col)
#pragma unroll
                for (int q = 0; q < PG_MR; ++q) {
                    const int r = part + RS * q;
                    if (r < M) s += fma(wcol[q].x, wcol[q].x, wcol[q].y * wcol[q].y);
                }
            const double bs = block_sum(s);
            if (tid == 0) st_agent(A.partn + blk, bs);
        }
        stamp(6);
        if (!pg_grid_sync(sy, bflag)) return;
        stamp(7);
        // ---- Hessenberg column j, Givens, residual: the same few operations in every workgroup -----
        if (wv == 0) {
            const double t = wave_sum(lane < A.G ? ld_agent(A.partn + lane) : 0.0);
            if (lane == 0) sc[3] = t;
        }
        __syncthreads();
        if (tid == 0) {
            const double hn = sqrt(fmax(sc[3], 0.0));
            cd* col = Hm + (size_t)j * (PG_RMAX + 1);
            for (int i = 0; i <= j; ++i) col[i] = cd{hs[i].x + hs[PS + i].x, hs[i].y + hs[PS + i].y};
            col[j + 1] = cd{hn, 0.0};
            for (int i = 0; i < j; ++i) {
                const cd a = col[i], bb = col[i + 1];
                const cd csc = cd{gcs[i].x, -gcs[i].y}, snc = cd{gsn[i].x, -gsn[i].y};
                const cd t1 = cmul(csc, a), t2 = cmul(snc, bb), t3 = cmul(gsn[i], a), t4 = cmul(gcs[i], bb);
                col[i] = cd{t1.x + t2.x, t1.y + t2.y};
                col[i + 1] = cd{-t3.x + t4.x, -t3.y + t4.y};
            }
            {
                const cd a = col[j], bb = col[j + 1];
                double den = hypot(hypot(a.x, a.y), hypot(bb.x, bb.y));
                if (den == 0.0) den = 1.0;
                gcs[j] = cd{a.x / den, a.y / den};
                gsn[j] = cd{bb.x / den, bb.y / den};
                col[j] = cd{den, 0.0};
                col[j + 1] = cd{0.0, 0.0};
                const cd gj = ggv[j];
                ggv[j] = cmul(cd{gcs[j].x, -gcs[j].y}, gj);
                const cd tmp = cmul(gsn[j], gj);
                ggv[j + 1] = cd{-tmp.x, -tmp.y};
            }
            const double rs = hypot(ggv[j + 1].x, ggv[j + 1].y) / bnorm;
            sc[0] = hn;
            sc[1] = rs;
            sc[2] = (rs <= A.tol || hn == 0.0) ? 1.0 : 0.0;
        }
        __syncthreads();
        iters = j + 1;
        resid = sc[1];
        const double hn = sc[0];
        if (sc[2] != 0.0) {
            converged = true;
            break;
        }
        {
            const double s = 1.0 / hn;
#pragma unroll
            for (int q = 0; q < PG_MR; ++q) vv[q] = cd{s * wcol[q].x, s * wcol[q].y};
        }
        __syncthreads();      // (sc is rewritten by the next iteration)
    }
    // ---- y from the triangular system, x = Kinv (V y) -------------------------------------------
    const int m = iters;
    if (tid == 0) {
        for (int i = m - 1; i >= 0; --i) {
            cd s = ggv[i];
            for (int k = i + 1; k < m; ++k) {
                const cd t = cmul(Hm[(size_t)k * (PG_RMAX + 1) + i], gy[k]);
                s.x -= t.x;
                s.y -= t.y;
            }
            const cd d = Hm[(size_t)i * (PG_RMAX + 1) + i];
            const double dd = d.x * d.x + d.y * d.y;
            gy[i] = dd == 0.0 ? cd{0.0, 0.0} : cd{(s.x * d.x + s.y * d.y) / dd, (s.y * d.x - s.x * d.y) / dd};
        }
    }
    __syncthreads();
    cd tcol[PG_MR], xr[PG_MR];
#pragma unroll
    for (int q = 0; q < PG_MR; ++q) tcol[q] = cd{0.0, 0.0};
    if (has_col) {
        for (int i = 0; i < m; ++i) {
            const cd ci = gy[i];
            const cd* Vi = A.V + (size_t)i * NB + c;
#pragma unroll
            for (int q = 0; q < PG_MR; ++q) {
                const int r = part + RS * q;
                if (r < M) {
                    const cd v = Vi[(size_t)r * n];
                    tcol[q].x += ci.x * v.x - ci.y * v.y;
                    tcol[q].y += ci.x * v.y + ci.y * v.x;
                }
            }
        }
    }
    precondition(tcol, xr);
    if (has_col) {
#pragma unroll
        for (int q = 0; q < PG_MR; ++q) {
            const int r = part + RS * q;
            if (r < M) A.x[(size_t)r * n + c] = xr[q];
        }
    }
    if (blk == 0 && tid == 0) {
        A.result[0] = (double)iters;
        A.result[1] = resid;
        A.result[2] = converged ? 1.0 : 0.0;
        A.result[3] = bnorm;
        for (int k = 0; k < 8; ++k) A.result[8 + k] = stage_ticks[k];
    }
}
