// K3 -- bundle-adjustment reprojection residual and analytic Jacobian blocks (gfx950).
//
// Replaces Optimizer.fun (scripts/lib/optimizer.py:174-229): for every observation
//   body2ned = quaternion_matrix(q)            lib/archive/transformations.py:1395-1420
//   R = body2cam . body2ned^T, t = -R.ned      lib/optimizer.py:120-126
//   (u,v) = pinhole + Brown(k1,k2,p1,p2,k3)    lib/project.py:300-329 (= cv2.projectPoints)
//   r = observed - projected                   lib/optimizer.py:222
// One thread per observation, float64 throughout; HBM-bound streaming of the
// camera-major observation list (idx 8 B + uv 16 B + point 24 B gathered + r 16 B = 64 B/obs,
// SURVEY.md 8d), camera blocks (56 B) are shared by consecutive observations and stay in L2.
//
// With body2cam = [[0,1,0],[0,0,1],[1,0,0]]:  Xc = (y1, y2, y0),  y = M(q)^T (X-ned) / |q|^2,
// M the homogeneous (unnormalised) rotation matrix of q = (w,x,y,z).
#include "iamx_common.h"
#include <cstdlib>

namespace {

constexpr double QEPS = 2.220446049250313e-16 * 4.0;   // transformations._EPS

struct Proj {
    double u, v;
    // intermediates kept for the Jacobian
    double x, y, r2, rad, iz;
    double yb[3];      // body-frame point
    double dX[3];
    double q[4];
    double inv_n;
    bool degenerate;
};

__device__ __forceinline__ void project_obs(const double *__restrict__ cam,
                                            const double *__restrict__ X,
                                            const double *__restrict__ cal, Proj &P)
{
    const double w = cam[3], x = cam[4], y = cam[5], z = cam[6];
    const double a = X[0] - cam[0], b = X[1] - cam[1], c = X[2] - cam[2];
    P.dX[0] = a; P.dX[1] = b; P.dX[2] = c;
    P.q[0] = w; P.q[1] = x; P.q[2] = y; P.q[3] = z;
    const double n = w * w + x * x + y * y + z * z;
    double y0, y1, y2;
    if (n < QEPS) {
        P.degenerate = true;
        P.inv_n = 0.0;
        y0 = a; y1 = b; y2 = c;
    } else {
        P.degenerate = false;
        const double inv_n = 1.0 / n;
        P.inv_n = inv_n;
        const double ww = w * w, xx = x * x, yy = y * y, zz = z * z;
        const double xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
        y0 = ((ww + xx - yy - zz) * a + 2.0 * (xy + wz) * b + 2.0 * (xz - wy) * c) * inv_n;
        y1 = (2.0 * (xy - wz) * a + (ww - xx + yy - zz) * b + 2.0 * (yz + wx) * c) * inv_n;
        y2 = (2.0 * (xz + wy) * a + 2.0 * (yz - wx) * b + (ww - xx - yy + zz) * c) * inv_n;
    }
    P.yb[0] = y0; P.yb[1] = y1; P.yb[2] = y2;
    const double iz = 1.0 / y0;            // camera z = body x
    const double px = y1 * iz, py = y2 * iz;
    const double r2 = px * px + py * py;
    const double k1 = cal[4], k2 = cal[5], p1 = cal[6], p2 = cal[7], k3 = cal[8];
    const double rad = 1.0 + r2 * (k1 + r2 * (k2 + r2 * k3));
    const double xd = px * rad + 2.0 * p1 * px * py + p2 * (r2 + 2.0 * px * px);
    const double yd = py * rad + p1 * (r2 + 2.0 * py * py) + 2.0 * p2 * px * py;
    P.u = cal[0] * xd + cal[2];
    P.v = cal[1] * yd + cal[3];
    P.x = px; P.y = py; P.r2 = r2; P.rad = rad; P.iz = iz;
}

// per camera: R (ned -> camera, row major) and the camera position; 12 doubles
__device__ __forceinline__ void cam_block(const double *__restrict__ cam, double *__restrict__ o)
{
    const double w = cam[3], x = cam[4], y = cam[5], z = cam[6];
    const double n = w * w + x * x + y * y + z * z;
    // rows of M(q)^T / n = body-frame axes; camera (X,Y,Z) = body (y1, y2, y0)
    if (n < QEPS) {
        o[0] = 1; o[1] = 0; o[2] = 0; o[3] = 0; o[4] = 1; o[5] = 0; o[6] = 0; o[7] = 0; o[8] = 1;
    } else {
        const double inv_n = 1.0 / n;
        const double ww = w * w, xx = x * x, yy = y * y, zz = z * z;
        const double xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
        o[0] = (ww + xx - yy - zz) * inv_n; o[1] = 2.0 * (xy + wz) * inv_n; o[2] = 2.0 * (xz - wy) * inv_n;
        o[3] = 2.0 * (xy - wz) * inv_n; o[4] = (ww - xx + yy - zz) * inv_n; o[5] = 2.0 * (yz + wx) * inv_n;
        o[6] = 2.0 * (xz + wy) * inv_n; o[7] = 2.0 * (yz - wx) * inv_n; o[8] = (ww - xx - yy + zz) * inv_n;
    }
    o[9] = cam[0]; o[10] = cam[1]; o[11] = cam[2];
}

// residual from the prepared camera blocks: the per-observation work is 9 FMAs, one
// reciprocal and the distortion polynomial; identical arithmetic to project_obs()
__device__ __forceinline__ double2 residual_rt(const double *__restrict__ R,
                                               const double *__restrict__ X, double2 obs,
                                               const double (&cal)[9])
{
    const double a = X[0] - R[9], b = X[1] - R[10], c = X[2] - R[11];
    const double y0 = (R[0] * a + R[1] * b + R[2] * c);
    const double y1 = (R[3] * a + R[4] * b + R[5] * c);
    const double y2 = (R[6] * a + R[7] * b + R[8] * c);
    const double iz = 1.0 / y0;
    const double px = y1 * iz, py = y2 * iz;
    const double r2 = px * px + py * py;
    const double rad = 1.0 + r2 * (cal[4] + r2 * (cal[5] + r2 * cal[8]));
    const double xd = px * rad + 2.0 * cal[6] * px * py + cal[7] * (r2 + 2.0 * px * px);
    const double yd = py * rad + cal[6] * (r2 + 2.0 * py * py) + 2.0 * cal[7] * px * py;
    return make_double2(obs.x - (cal[0] * xd + cal[2]), obs.y - (cal[1] * yd + cal[3]));
}

// One launch: every WAVE (128 observations) first builds the camera blocks of the camera range
// it touches in its slice of LDS (camera-major observations: 1-2 cameras), then evaluates from
// LDS.  A wave that spans more than RES_MAXCAM cameras (any observation order is legal) builds
// the block per observation in registers instead.
constexpr int RES_MAXCAM = 16;

// Workgroups are dealt to the 8 XCDs round robin; consecutive blocks of camera-major observations
// gather neighbouring points.  Give every XCD a contiguous run of blocks (each has its own L2).
__device__ __forceinline__ int xcd_contiguous_block(int bid, int n)
{
    const int xcd = bid & 7, k = bid >> 3, q = n >> 3, r = n & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

__global__ __launch_bounds__(256) void ba_residual_lds_kernel(
    const double *__restrict__ cams, const double *__restrict__ pts,
    const int32_t *__restrict__ cam_idx, const int32_t *__restrict__ pt_idx,
    const double *__restrict__ uv, int64_t n_obs, const double *__restrict__ calib,
    double *__restrict__ r)
{
    __shared__ double Rs_all[4][RES_MAXCAM][12];        // one set of camera blocks per WAVE
    double (*Rs)[12] = Rs_all[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    const int64_t o = ((int64_t)xcd_contiguous_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x) * 2;
    const bool two = o + 1 < n_obs, one = o < n_obs;
    int2 ci = make_int2(0, 0), pi = make_int2(0, 0);
    double4 ob = make_double4(0, 0, 0, 0);
    // every load of the two observations is issued BEFORE the camera-block prologue (two
    // barriers, LDS atomics, a quaternion -> matrix conversion): the kernel is one dependent
    // chain per workgroup otherwise -- index, barrier, camera block, barrier, point gather,
    // arithmetic -- and with 125 MB served out of the Infinity Cache the latencies, not the
    // bytes, are the time (31 -> 27.5 us per launch under rocprofv3; four observations per
    // thread: 35.8 us, the 64-byte per-thread strides cost more than the extra loads in flight)
    if (two) {
        ci = *reinterpret_cast<const int2 *>(cam_idx + o);
        pi = *reinterpret_cast<const int2 *>(pt_idx + o);
        ob = *reinterpret_cast<const double4 *>(uv + 2 * o);
    } else if (one) {
        ci.x = ci.y = cam_idx[o];
        pi.x = pi.y = pt_idx[o];
        const double2 t = *reinterpret_cast<const double2 *>(uv + 2 * o);
        ob = make_double4(t.x, t.y, 0, 0);
    }
    double X0[3], X1[3];
    {
        const double *q0 = pts + (int64_t)pi.x * 3, *q1 = pts + (int64_t)pi.y * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) { X0[k] = q0[k]; X1[k] = q1[k]; }
    }
    // camera range of the workgroup (any observation order is handled; camera-major makes it small)
    int lo = one ? min(ci.x, ci.y) : 0x7FFFFFFF, hi = one ? max(ci.x, ci.y) : -1;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        lo = min(lo, __shfl_xor(lo, m));
        hi = max(hi, __shfl_xor(hi, m));
    }
    // (per wave, wave-synchronous: no workgroup barrier anywhere in this kernel; a wave whose
    //  128 observations are all past the end has hi = -1 and builds nothing)
    const int c_lo = lo, ncam = hi - lo + 1;
    const bool in_lds = ncam <= RES_MAXCAM;
    if (in_lds && lane < ncam) cam_block(cams + (int64_t)(c_lo + lane) * 7, Rs[lane]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (!one) return;
    double cal[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) cal[i] = calib[i];
    double Rl0[12], Rl1[12];
    if (!in_lds) {
        cam_block(cams + (int64_t)ci.x * 7, Rl0);
        cam_block(cams + (int64_t)ci.y * 7, Rl1);
    }
    const double *R0 = in_lds ? Rs[ci.x - c_lo] : Rl0;
    const double2 r0 = residual_rt(R0, X0, make_double2(ob.x, ob.y), cal);
    if (two) {
        const double *R1 = in_lds ? Rs[ci.y - c_lo] : Rl1;
        const double2 r1 = residual_rt(R1, X1, make_double2(ob.z, ob.w), cal);
        *reinterpret_cast<double4 *>(r + 2 * o) = make_double4(r0.x, r0.y, r1.x, r1.y);
    } else {
        *reinterpret_cast<double2 *>(r + 2 * o) = r0;
    }
}

// The same evaluation as a PERSISTENT, software-pipelined kernel.  ba_residual_lds_kernel is one
// dependent chain per workgroup -- indices, then the point gather and the camera parameters behind
// them, then arithmetic, then the store -- and a workgroup lives for exactly one such chain: with
// several generations of workgroups per launch the two memory latencies, not the 64 B per
// observation, are the time (profiles/r5: 92 MB of HBM traffic in 27 us).  Here every WAVE walks
// its share of the 128-observation chunks with three of them in flight: while chunk k is being
// evaluated the point gather + camera parameters of chunk k+1 (whose indices arrived during the
// previous step) and the indices / observed pixels of chunk k+2 are on their way.
// The loop body has NO branch around a load (clamped addresses instead): the compiler's wait
// counters then stay exact and a step waits for the loads it needs, not for all of them (a first
// version with `if (in range) load` waited for vmcnt(0) every step and ran slower than the form it
// was to replace, profiles/r6_ba_resid_ab.txt).  Wave synchronous, no workgroup barrier; camera
// blocks per wave in LDS, double buffered.  A chunk that spans more than PIPE_MAXCAM cameras
// (legal: any observation order is) is remembered and redone per observation AFTER the loop.
constexpr int PIPE_MAXCAM = 8;

struct ObsIdx {
    int2 ci, pi;
    double4 ob;
};

template <int PIPE_SETS>
__global__ __launch_bounds__(256) void ba_residual_pipe_kernel(
    const double *__restrict__ cams, const double *__restrict__ pts,
    const int32_t *__restrict__ cam_idx, const int32_t *__restrict__ pt_idx,
    const double *__restrict__ uv, int64_t n_obs, const double *__restrict__ calib,
    double *__restrict__ r, int steps, int n_cams)
{
    __shared__ double Rs_all[4][2][PIPE_MAXCAM][12];    // per WAVE, double buffered
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t n_chunks = (n_obs + 127) / 128;
    // (workgroups are dealt to the XCDs round robin; inside one step of the walk every XCD gets a
    //  contiguous run of chunks: neighbouring chunks gather neighbouring points, one L2 each)
    const int64_t wg = xcd_contiguous_block(blockIdx.x, gridDim.x);
    const int64_t first = wg * 4 + wave, stride = (int64_t)gridDim.x * 4;
    if (first >= n_chunks) return;
    double cal[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) cal[i] = calib[i];
    const int64_t last_pair = n_obs - 2;                 // (n_obs >= 2: the launcher checks)
    // A lane's observation pair of chunk c is ALWAYS a valid pair: chunks past the end repeat the
    // last chunk, pairs past the end repeat the last pair (n_obs - 2, n_obs - 1) -- such a lane
    // computes and stores the same values to the same place as the lane that owns them.  So no
    // load and no store of the walk sits behind a branch.
    auto pair_of = [&](int64_t c) {
        const int64_t cc = c < n_chunks ? c : n_chunks - 1;
        const int64_t o = (cc * 64 + lane) * 2;
        return o < last_pair ? o : last_pair;
    };
    auto load_idx = [&](int64_t c, ObsIdx &I) {
        const int64_t ol = pair_of(c);
        I.ci = *reinterpret_cast<const int2 *>(cam_idx + ol);
        I.pi = *reinterpret_cast<const int2 *>(pt_idx + ol);
        I.ob = *reinterpret_cast<const double4 *>(uv + 2 * ol);
    };
    // the gather stage of a chunk whose indices have arrived: both points, the camera range of
    // the wave, the parameters of camera lo + min(lane, PIPE_MAXCAM - 1)
    auto gather = [&](const ObsIdx &I, double (&x0)[3], double (&x1)[3], double (&cpar)[7], int &lo_out,
                      int &hi_out) {
        const double *q0 = pts + (int64_t)I.pi.x * 3, *q1 = pts + (int64_t)I.pi.y * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) { x0[k] = q0[k]; x1[k] = q1[k]; }
        // camera range of the chunk: observations are camera-major in practice, so the first
        // lane's first camera and the last lane's second bound it (two v_readlane instead of two
        // dependent chains of six cross-lane exchanges, which one wave per SIMD cannot hide); a
        // chunk where some lane falls outside gets hi = lo + PIPE_MAXCAM: the redo path
        int lo = __builtin_amdgcn_readlane(I.ci.x, 0), hi = __builtin_amdgcn_readlane(I.ci.y, 63);
        const bool inside = I.ci.x >= lo && I.ci.x <= hi && I.ci.y >= lo && I.ci.y <= hi;
        if (__ballot(inside) != ~0ull || hi < lo) hi = lo + PIPE_MAXCAM;
        lo_out = lo;
        hi_out = hi;
        const int cl = lo + (lane < PIPE_MAXCAM ? lane : PIPE_MAXCAM - 1);
        const double *cs = cams + (int64_t)min(min(cl, hi), n_cams - 1) * 7;
#pragma unroll
        for (int k = 0; k < 7; ++k) cpar[k] = cs[k];
    };
    // PIPE_SETS register sets take the roles "indices on their way" (chunks k+4, k+3), "gather on
    // its way" (k+2, k+1) and "being evaluated" (k) in turn: with one wave per SIMD a step is as
    // long as the slowest dependent load it waits for, and two steps of slack per stage halve
    // that (three sets, one step of slack: 27.5 us on a rotating working set; profiles/r6).  The
    // walk is unrolled by PIPE_SETS so that the roles rotate by NAME -- a register copy of a set
    // whose loads are still in flight would make the step wait for them.  The launcher sizes the
    // grid so that the steps per wave are a multiple of PIPE_SETS (`steps`).
    struct Set {
        ObsIdx idx;
        double X0[3], X1[3], cp[7];
        int lo, hi;
    };
    Set S[PIPE_SETS];
    unsigned long long redo = 0ull;     // steps whose chunk spans more than PIPE_MAXCAM cameras
    int step = 0;
    auto do_step = [&](Set &Cs, Set &Gs, Set &As, int64_t c) __attribute__((always_inline)) {
        load_idx(c + (PIPE_SETS - 1) * stride, As.idx);                       // chunk k+4 (k+2): indices
        gather(Gs.idx, Gs.X0, Gs.X1, Gs.cp, Gs.lo, Gs.hi);                    // chunk k+2: gather
        // chunk k -- camera blocks into this wave's LDS slice, then the residuals
        double (*Rs)[12] = Rs_all[wave][step & 1];
        if (lane < PIPE_MAXCAM) cam_block(Cs.cp, Rs[lane]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        redo |= (Cs.hi - Cs.lo >= PIPE_MAXCAM && c < n_chunks) ? 1ull << step : 0ull;    // (wave uniform)
        const int k0 = max(0, min(Cs.idx.ci.x - Cs.lo, PIPE_MAXCAM - 1)), k1 = max(0, min(Cs.idx.ci.y - Cs.lo, PIPE_MAXCAM - 1));
        const double2 r0 = residual_rt(Rs[k0], Cs.X0, make_double2(Cs.idx.ob.x, Cs.idx.ob.y), cal);
        const double2 r1 = residual_rt(Rs[k1], Cs.X1, make_double2(Cs.idx.ob.z, Cs.idx.ob.w), cal);
        *reinterpret_cast<double4 *>(r + 2 * pair_of(c)) = make_double4(r0.x, r0.y, r1.x, r1.y);
        ++step;
    };
    constexpr int GA = (PIPE_SETS - 1) / 2;                // the gather runs GA steps ahead
#pragma unroll
    for (int t = 0; t < PIPE_SETS - 1; ++t) load_idx(first + t * stride, S[t].idx);
#pragma unroll
    for (int t = 0; t < GA; ++t) gather(S[t].idx, S[t].X0, S[t].X1, S[t].cp, S[t].lo, S[t].hi);
    {
        int64_t c = first;
        for (int g = 0; g < steps; g += PIPE_SETS) {
#pragma unroll
            for (int j = 0; j < PIPE_SETS; ++j) {
                do_step(S[j], S[(j + GA) % PIPE_SETS], S[(j + PIPE_SETS - 1) % PIPE_SETS], c);
                c += stride;
            }
        }
    }
    // the chunks whose camera range did not fit the LDS slice: per observation, from registers
    while (redo) {
        const int st = __builtin_ctzll(redo);
        redo &= redo - 1;
        const int64_t o = ((first + (int64_t)st * stride) * 64 + lane) * 2;
        for (int h = 0; h < 2; ++h) {
            if (o + h >= n_obs) break;
            double Rl[12], X[3];
            cam_block(cams + (int64_t)cam_idx[o + h] * 7, Rl);
            const double *q = pts + (int64_t)pt_idx[o + h] * 3;
            X[0] = q[0]; X[1] = q[1]; X[2] = q[2];
            const double2 ob = *reinterpret_cast<const double2 *>(uv + 2 * (o + h));
            *reinterpret_cast<double2 *>(r + 2 * (o + h)) = residual_rt(Rl, X, ob, cal);
        }
    }
}

__global__ __launch_bounds__(256) void ba_residual_kernel(
    const double *__restrict__ cams, int n_cams, const double *__restrict__ pts,
    const int32_t *__restrict__ cam_idx, const int32_t *__restrict__ pt_idx,
    const double *__restrict__ uv, int64_t n_obs, const double *__restrict__ calib,
    double *__restrict__ r)
{
    double cal[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) cal[i] = calib[i];
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < n_obs;
         o += (int64_t)gridDim.x * 256) {
        const int ci = cam_idx[o];
        const int pi = pt_idx[o];
        const double2 obs = *reinterpret_cast<const double2 *>(uv + 2 * o);
        Proj P;
        project_obs(cams + (int64_t)ci * 7, pts + (int64_t)pi * 3, cal, P);
        *reinterpret_cast<double2 *>(r + 2 * o) = make_double2(obs.x - P.u, obs.y - P.v);
    }
}

template <bool WITH_CALIB>
__global__ __launch_bounds__(256) void ba_residual_jac_kernel(
    const double *__restrict__ cams, int n_cams, const double *__restrict__ pts,
    const int32_t *__restrict__ cam_idx, const int32_t *__restrict__ pt_idx,
    const double *__restrict__ uv, int64_t n_obs, const double *__restrict__ calib,
    double *__restrict__ r, double *__restrict__ Jc, double *__restrict__ Jp,
    double *__restrict__ Jk)
{
    double cal[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) cal[i] = calib[i];
    const double fx = cal[0], fy = cal[1];
    const double k1 = cal[4], k2 = cal[5], p1 = cal[6], p2 = cal[7], k3 = cal[8];
    // Without calibration columns the 14 + 6 doubles of an observation are staged in LDS and
    // the workgroup's contiguous [256][14] / [256][6] output ranges are written with coalesced
    // stores (a thread writing its own 112-byte record makes every store instruction touch 64
    // different cache lines).
    constexpr int SC = WITH_CALIB ? 1 : 15, SP = WITH_CALIB ? 1 : 7;      // padded LDS strides
    __shared__ double sJc[WITH_CALIB ? 1 : 256 * SC];
    __shared__ double sJp[WITH_CALIB ? 1 : 256 * SP];
    // (the XCD-contiguous block order of the residual kernel measured 3 % slower here: the grid is
    //  capped and strided, and the kernel is bound by its 224 B/observation of output)
    for (int64_t base = (int64_t)blockIdx.x * 256; base < n_obs; base += (int64_t)gridDim.x * 256) {
        const int64_t o = base + threadIdx.x;
        const bool valid = o < n_obs;
        if (valid) {
        const int ci = cam_idx[o];
        const int pi = pt_idx[o];
        Proj P;
        project_obs(cams + (int64_t)ci * 7, pts + (int64_t)pi * 3, cal, P);
        if (r) {
            const double2 obs = *reinterpret_cast<const double2 *>(uv + 2 * o);
            *reinterpret_cast<double2 *>(r + 2 * o) = make_double2(obs.x - P.u, obs.y - P.v);
        }
        const double x = P.x, y = P.y, r2 = P.r2, rad = P.rad, iz = P.iz;
        const double drad = k1 + r2 * (2.0 * k2 + 3.0 * k3 * r2);
        // d(xd,yd)/d(x,y)
        const double xdx = rad + 2.0 * x * x * drad + 2.0 * p1 * y + 6.0 * p2 * x;
        const double xdy = 2.0 * x * y * drad + 2.0 * p1 * x + 2.0 * p2 * y;
        const double ydx = xdy;
        const double ydy = rad + 2.0 * y * y * drad + 6.0 * p1 * y + 2.0 * p2 * x;
        // d(u,v)/d(body point yb): camera (X,Y,Z) = (yb1, yb2, yb0)
        //   x = yb1/yb0, y = yb2/yb0
        const double ux = fx * xdx, uy = fx * xdy, vx = fy * ydx, vy = fy * ydy;
        double du[3], dv[3];
        du[0] = -(ux * x + uy * y) * iz;   dv[0] = -(vx * x + vy * y) * iz;
        du[1] = ux * iz;                    dv[1] = vx * iz;
        du[2] = uy * iz;                    dv[2] = vy * iz;

        // d yb / d X = B^T (rows of M^T / n);  d yb / d ned = -B^T
        const double w = P.q[0], qx = P.q[1], qy = P.q[2], qz = P.q[3];
        const double a = P.dX[0], b = P.dX[1], c = P.dX[2];
        const double inv_n = P.inv_n;
        double BT[3][3];
        if (P.degenerate) {
            BT[0][0] = 1; BT[0][1] = 0; BT[0][2] = 0;
            BT[1][0] = 0; BT[1][1] = 1; BT[1][2] = 0;
            BT[2][0] = 0; BT[2][1] = 0; BT[2][2] = 1;
        } else {
            const double ww = w * w, xx = qx * qx, yy = qy * qy, zz = qz * qz;
            const double xy = qx * qy, xz = qx * qz, yz = qy * qz;
            const double wx = w * qx, wy = w * qy, wz = w * qz;
            BT[0][0] = (ww + xx - yy - zz) * inv_n; BT[0][1] = 2.0 * (xy + wz) * inv_n; BT[0][2] = 2.0 * (xz - wy) * inv_n;
            BT[1][0] = 2.0 * (xy - wz) * inv_n; BT[1][1] = (ww - xx + yy - zz) * inv_n; BT[1][2] = 2.0 * (yz + wx) * inv_n;
            BT[2][0] = 2.0 * (xz + wy) * inv_n; BT[2][1] = 2.0 * (yz - wx) * inv_n; BT[2][2] = (ww - xx - yy + zz) * inv_n;
        }
        double jp_u[3], jp_v[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            jp_u[j] = du[0] * BT[0][j] + du[1] * BT[1][j] + du[2] * BT[2][j];
            jp_v[j] = dv[0] * BT[0][j] + dv[1] * BT[1][j] + dv[2] * BT[2][j];
        }
        // d yb / d q_k = (dM^T/dq_k . dX - 2 q_k yb) / n
        double jq_u[4], jq_v[4];
        if (P.degenerate) {
#pragma unroll
            for (int k = 0; k < 4; ++k) jq_u[k] = jq_v[k] = 0.0;
        } else {
            const double s = w * a + qz * b - qy * c;     // recurring bilinear forms
            const double t = -qz * a + w * b + qx * c;
            const double p = qy * a - qx * b + w * c;
            const double d = qx * a + qy * b + qz * c;
            const double dvq[4][3] = {{s, t, p}, {d, p, -t}, {-p, d, s}, {t, -s, d}};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double qk = P.q[k];
                const double e0 = 2.0 * (dvq[k][0] - qk * P.yb[0]) * inv_n;
                const double e1 = 2.0 * (dvq[k][1] - qk * P.yb[1]) * inv_n;
                const double e2 = 2.0 * (dvq[k][2] - qk * P.yb[2]) * inv_n;
                jq_u[k] = du[0] * e0 + du[1] * e1 + du[2] * e2;
                jq_v[k] = dv[0] * e0 + dv[1] * e1 + dv[2] * e2;
            }
        }
        // r = observed - projected  =>  J = -d(u,v)/d(params)
        double *jc = WITH_CALIB ? Jc + o * 14 : sJc + threadIdx.x * SC;
        jc[0] = jp_u[0]; jc[1] = jp_u[1]; jc[2] = jp_u[2];          // d/d ned = -(-B^T) -> +
        jc[3] = -jq_u[0]; jc[4] = -jq_u[1]; jc[5] = -jq_u[2]; jc[6] = -jq_u[3];
        jc[7] = jp_v[0]; jc[8] = jp_v[1]; jc[9] = jp_v[2];
        jc[10] = -jq_v[0]; jc[11] = -jq_v[1]; jc[12] = -jq_v[2]; jc[13] = -jq_v[3];
        double *jp = WITH_CALIB ? Jp + o * 6 : sJp + threadIdx.x * SP;
        jp[0] = -jp_u[0]; jp[1] = -jp_u[1]; jp[2] = -jp_u[2];
        jp[3] = -jp_v[0]; jp[4] = -jp_v[1]; jp[5] = -jp_v[2];
        if constexpr (WITH_CALIB) {
            const double xd = (P.u - cal[2]) / fx, yd = (P.v - cal[3]) / fy;
            const double r4 = r2 * r2, r6 = r4 * r2;
            double *jk = Jk + o * 16;
            // (f, cu, cv, k1, k2, p1, p2, k3), fx = fy = f
            jk[0] = -xd;  jk[1] = -1.0; jk[2] = 0.0;
            jk[3] = -fx * x * r2; jk[4] = -fx * x * r4;
            jk[5] = -fx * 2.0 * x * y; jk[6] = -fx * (r2 + 2.0 * x * x); jk[7] = -fx * x * r6;
            jk[8] = -yd;  jk[9] = 0.0; jk[10] = -1.0;
            jk[11] = -fy * y * r2; jk[12] = -fy * y * r4;
            jk[13] = -fy * (r2 + 2.0 * y * y); jk[14] = -fy * 2.0 * x * y; jk[15] = -fy * y * r6;
        }
        }   // valid
        if constexpr (!WITH_CALIB) {
            __syncthreads();
            const int cnt = (int)((n_obs - base) < 256 ? (n_obs - base) : 256);
            for (int e = threadIdx.x; e < cnt * 14; e += 256) Jc[base * 14 + e] = sJc[(e / 14) * SC + e % 14];
            for (int e = threadIdx.x; e < cnt * 6; e += 256) Jp[base * 6 + e] = sJp[(e / 6) * SP + e % 6];
            __syncthreads();
        }
    }
}

inline unsigned grid_for(int64_t n)
{
    int64_t g = (n + 255) / 256;
    const int64_t cap = 256 * 16;      // 256 CUs x 16 resident blocks is plenty; grid-stride the rest
    return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int iamx_ba_residual(const double *cams, int n_cams, const double *pts, int n_pts,
                                const int32_t *cam_idx, const int32_t *pt_idx, const double *uv,
                                int64_t n_obs, const double *calib, double *r, void *stream)
{
    IAMX_REQUIRE(cams && pts && cam_idx && pt_idx && uv && calib && r, "null pointer");
    IAMX_REQUIRE(n_cams > 0 && n_pts > 0 && n_obs >= 0, "bad size");
    if (n_obs == 0) return IAMX_OK;
    hipLaunchKernelGGL(ba_residual_kernel, dim3(grid_for(n_obs)), dim3(256), 0,
                       iamx::as_stream(stream), cams, n_cams, pts, cam_idx, pt_idx, uv, n_obs,
                       calib, r);
    return iamx::check_launch("iamx_ba_residual");
}

extern "C" int iamx_ba_residual_prepared(const double *cams, int n_cams, const double *pts,
                                         int n_pts, const int32_t *cam_idx,
                                         const int32_t *pt_idx, const double *uv, int64_t n_obs,
                                         const double *calib, double *cam_scratch, double *r,
                                         void *stream)
{
    IAMX_REQUIRE(cams && pts && cam_idx && pt_idx && uv && calib && r, "null pointer");
    IAMX_REQUIRE(n_cams > 0 && n_pts > 0 && n_obs >= 0, "bad size");
    if (n_obs == 0) return IAMX_OK;
    IAMX_REQUIRE((((uintptr_t)uv | (uintptr_t)r) & 31) == 0 &&
                     (((uintptr_t)cam_idx | (uintptr_t)pt_idx) & 7) == 0,
                 "uv / r must be 32-byte aligned, cam_idx / pt_idx 8-byte aligned");
    hipStream_t st = iamx::as_stream(stream);
    (void)cam_scratch;           // kept in the signature; the camera blocks now live in LDS
    // IAMX_BA_RESIDUAL=lds: the one-chain-per-workgroup form (A/B: tools/ba_resid_ab.py);
    // default: the persistent pipelined walk on IAMX_BA_RESIDUAL_WGS workgroups -- at least so many
    // that no wave walks more than 64 chunks (the redo mask), never more than there are chunks
    const char *form = getenv("IAMX_BA_RESIDUAL"), *wgs = getenv("IAMX_BA_RESIDUAL_WGS");
    if ((form && form[0] == 'l') || n_obs < 2) {
        hipLaunchKernelGGL(ba_residual_lds_kernel, dim3((unsigned)((n_obs + 511) / 512)), dim3(256), 0,
                           st, cams, pts, cam_idx, pt_idx, uv, n_obs, calib, r);
    } else {
        // steps per wave: a multiple of PIPE_SETS (the walk is unrolled by it), at most 60 (the
        // redo mask); then as few workgroups as cover the chunks in that many steps -- 256 (one per
        // CU, one wave per SIMD) x 15 steps at configs[3]
        const int64_t n_chunks = (n_obs + 127) / 128;
        const char *sets_e = getenv("IAMX_BA_RESIDUAL_SETS");
        const int sets = sets_e && sets_e[0] == '5' ? 5 : 3;
        int64_t g = wgs ? atoll(wgs) : 256;
        if (g < 1) g = 1;
        int64_t steps = (n_chunks + 4 * g - 1) / (4 * g);
        steps = (steps + sets - 1) / sets * sets;
        if (steps > 60) steps = 60;
        g = (n_chunks + 4 * steps - 1) / (4 * steps);
        if (sets == 5)
            hipLaunchKernelGGL(ba_residual_pipe_kernel<5>, dim3((unsigned)g), dim3(256), 0, st, cams, pts,
                               cam_idx, pt_idx, uv, n_obs, calib, r, (int)steps, n_cams);
        else
            hipLaunchKernelGGL(ba_residual_pipe_kernel<3>, dim3((unsigned)g), dim3(256), 0, st, cams, pts,
                               cam_idx, pt_idx, uv, n_obs, calib, r, (int)steps, n_cams);
    }
    return iamx::check_launch("iamx_ba_residual_prepared");
}

extern "C" int iamx_ba_residual_jac(const double *cams, int n_cams, const double *pts, int n_pts,
                                    const int32_t *cam_idx, const int32_t *pt_idx,
                                    const double *uv, int64_t n_obs, const double *calib,
                                    double *r, double *Jc, double *Jp, double *Jk, void *stream)
{
    IAMX_REQUIRE(cams && pts && cam_idx && pt_idx && uv && calib && Jc && Jp, "null pointer");
    IAMX_REQUIRE(n_cams > 0 && n_pts > 0 && n_obs >= 0, "bad size");
    if (n_obs == 0) return IAMX_OK;
    if (Jk)
        hipLaunchKernelGGL(ba_residual_jac_kernel<true>, dim3(grid_for(n_obs)), dim3(256), 0,
                           iamx::as_stream(stream), cams, n_cams, pts, cam_idx, pt_idx, uv, n_obs,
                           calib, r, Jc, Jp, Jk);
    else
        hipLaunchKernelGGL(ba_residual_jac_kernel<false>, dim3(grid_for(n_obs)), dim3(256), 0,
                           iamx::as_stream(stream), cams, n_cams, pts, cam_idx, pt_idx, uv, n_obs,
                           calib, r, Jc, Jp, Jk);
    return iamx::check_launch("iamx_ba_residual_jac");
}

// The yardstick of the HBM-bound kernels above: a plain grid-stride copy, 16 bytes per lane and
// step (the form MI355X_MICROARCH.md measures at 6.29 TB/s), over whatever working set the caller
// rotates through.  n16 = number of 16-byte words WRITTEN; reads_per_write source words are read
// (and added up) per word written: 1 = a copy, 3 = the read : write mix of the residual kernel
// (48 B in, 16 B out per observation).
namespace {
template <int K>
__global__ __launch_bounds__(256) void copy16_kernel(const double2 *__restrict__ src,
                                                     double2 *__restrict__ dst, int64_t n16)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
        double2 acc = src[i];
#pragma unroll
        for (int k = 1; k < K; ++k) {
            const double2 v = src[(int64_t)k * n16 + i];
            acc.x += v.x;
            acc.y += v.y;
        }
        dst[i] = acc;
    }
}
}  // namespace

extern "C" int iamx_hbm_copy16(const void *src, void *dst, int64_t n16, int reads_per_write,
                               int workgroups, void *stream)
{
    IAMX_REQUIRE(src && dst && n16 >= 0 && workgroups > 0, "bad argument");
    IAMX_REQUIRE(reads_per_write >= 1 && reads_per_write <= 4, "reads_per_write must be 1..4");
    IAMX_REQUIRE((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "16-byte alignment");
    if (n16 == 0) return IAMX_OK;
    const dim3 g((unsigned)workgroups), b(256);
    hipStream_t st = iamx::as_stream(stream);
    const double2 *s2 = reinterpret_cast<const double2 *>(src);
    double2 *d2 = reinterpret_cast<double2 *>(dst);
    switch (reads_per_write) {
    case 1: hipLaunchKernelGGL(copy16_kernel<1>, g, b, 0, st, s2, d2, n16); break;
    case 2: hipLaunchKernelGGL(copy16_kernel<2>, g, b, 0, st, s2, d2, n16); break;
    case 3: hipLaunchKernelGGL(copy16_kernel<3>, g, b, 0, st, s2, d2, n16); break;
    default: hipLaunchKernelGGL(copy16_kernel<4>, g, b, 0, st, s2, d2, n16); break;
    }
    return iamx::check_launch("iamx_hbm_copy16");
}
