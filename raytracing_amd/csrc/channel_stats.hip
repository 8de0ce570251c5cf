// Channel statistics of the result table, per receiver: RMS delay spread, RMS angular spreads, share of the power in
// the n strongest rays and the angular window holding a given share of the power
// (electromagnetism_fun/electromagnetism_plots.py:287-460).  One wave per receiver; its paths are contiguous.
//
// The three RMS figures and the share are sums over the receiver's paths (lanes stride over them, shuffles reduce).
// The angular window is a selection: paths are gathered in order of their distance on the unit sphere from a centre
// path until the share is reached, then the two gathered paths furthest apart give the window.  Which paths are picked
// decides the result, so those distances are computed with the reference's roundings (products and sums that are
// never fused) and ties go to the lowest index as numpy.argmin / argmax do.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <string>

#include "../../include/rt_pathsolve.h"
#include "rt_common.h"

namespace {

__device__ __forceinline__ double wave_sum_f64(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ double wave_min_f64(double v) {
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off));
    return v;
}
// best (value, index) pair over the wave: larger value wins when LARGEST (smaller otherwise), ties to the lower index
template <bool LARGEST>
__device__ __forceinline__ void wave_arg(double &v, int &i) {
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(v, off);
        const int oi = __shfl_xor(i, off);
        const bool better = oi >= 0 && (i < 0 || (LARGEST ? ov > v : ov < v) || (ov == v && oi < i));
        if (better) {
            v = ov;
            i = oi;
        }
    }
}

// electromagnetism_utils.py:38-42
__device__ __forceinline__ double angle_distance(double alpha, double beta) {
    const double phi = fmod(fabs(beta - alpha), 360.0);
    return phi > 180.0 ? 360.0 - phi : phi;
}

// sqrt(sum((a - b)**2)) with numpy's roundings (electromagnetism_plots.py:316, 336)
__device__ __forceinline__ double sphere_distance(const double *a, const double *b) {
    const double dx = __dsub_rn(a[0], b[0]), dy = __dsub_rn(a[1], b[1]), dz = __dsub_rn(a[2], b[2]);
    return __dsqrt_rn(__dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz)));
}

struct StatsIn {
    const int64_t *off;
    const double *power, *time, *el, *az, *xyz;
    const uint8_t *is_los;
};

// out[g*8 ..]: 0 RMS delay spread, 1 RMS azimuth spread, 2 RMS elevation spread (degrees), 3 share of the n strongest
// rays (percent), 4 1.0 when the receiver has a line of sight, 5 window azimuth, 6 window elevation (degrees),
// 7 number of paths.  dist / order: scratch, one slot per path.
__global__ __launch_bounds__(64) void k_channel_stats(StatsIn in, int64_t n_groups, int n_strongest, double power_percent,
                                                      double *__restrict__ dist, int32_t *__restrict__ order,
                                                      double *__restrict__ out) {
    const int64_t g = blockIdx.x;
    if (g >= n_groups) return;
    const int lane = threadIdx.x;
    const int64_t p0 = in.off[g];
    const int P = (int)(in.off[g + 1] - p0);
    double *o = out + 8 * g;
    if (P <= 0) {
        if (lane < 8) o[lane] = lane == 7 ? 0.0 : NAN;
        return;
    }
    const double *pw = in.power + p0, *tm = in.time + p0, *el = in.el + p0, *az = in.az + p0, *xyz = in.xyz + 3 * p0;
    double *d = dist + p0;
    int32_t *ord = order + p0;

    // ---- sums: total power, earliest arrival, strongest ray (first of equals) ----
    double s_pow = 0, s_tp = 0, t_min = INFINITY, best = -INFINITY;
    int best_i = -1, los = 0;
    for (int k = lane; k < P; k += 64) {
        s_pow += pw[k];
        s_tp += tm[k] * pw[k];
        t_min = fmin(t_min, tm[k]);
        if (pw[k] > best) {
            best = pw[k];
            best_i = k;
        }
        los |= in.is_los[p0 + k];
    }
    const double total = wave_sum_f64(s_pow);
    const double t0 = wave_min_f64(t_min);
    wave_arg<true>(best, best_i);
    const int strongest = best_i;
    const bool has_los = __any(los != 0);
    // RMS delay spread (electromagnetism_plots.py:389-400)
    const double t_mean = wave_sum_f64(s_tp) / total - t0;
    // RMS angular spreads about the strongest ray (electromagnetism_plots.py:403-431)
    const double ref_el = el[strongest], ref_az = az[strongest];
    const double to_rad = M_PI / 180.0;
    double s_t2 = 0, s_e = 0, s_a = 0;
    for (int k = lane; k < P; k += 64) {
        const double dt = tm[k] - t0 - t_mean;
        s_t2 += dt * dt * pw[k];
        s_e += angle_distance(el[k], ref_el) * to_rad * pw[k];
        s_a += angle_distance(az[k], ref_az) * to_rad * pw[k];
    }
    const double delay = sqrt(wave_sum_f64(s_t2) / total);
    const double e_mean = wave_sum_f64(s_e) / total, a_mean = wave_sum_f64(s_a) / total;
    double s_e2 = 0, s_a2 = 0;
    for (int k = lane; k < P; k += 64) {
        const double de = angle_distance(el[k], ref_el) * to_rad - e_mean, da = angle_distance(az[k], ref_az) * to_rad - a_mean;
        s_e2 += de * de * pw[k];
        s_a2 += da * da * pw[k];
    }
    const double rms_el = sqrt(wave_sum_f64(s_e2) / total) * (180.0 / M_PI);
    const double rms_az = sqrt(wave_sum_f64(s_a2) / total) * (180.0 / M_PI);

    // ---- share of the n strongest rays (electromagnetism_plots.py:445-460): strongest first, n times ----
    for (int k = lane; k < P; k += 64) d[k] = pw[k];
    __builtin_amdgcn_wave_barrier();
    double share = 0;
    const int n_take = n_strongest < P ? n_strongest : P;
    for (int r = 0; r < n_take; ++r) {
        double v = -INFINITY;
        int vi = -1;
        for (int k = lane; k < P; k += 64)
            if (d[k] > v) {
                v = d[k];
                vi = k;
            }
        wave_arg<true>(v, vi);
        share += v;
        if (lane == 0) d[vi] = -INFINITY;
        __builtin_amdgcn_wave_barrier();
    }
    const double percent = share / total * 100.0;

    // ---- angular window (electromagnetism_plots.py:287-387) ----
    double win_el = 0, win_az = 0;
    const double top_share = pw[strongest] / total;
    if (!(top_share > power_percent)) {
        // The centre of the bundle: the strongest ray when it must belong to it; otherwise the reference tries every
        // ray as the centre and keeps the spread of each one it tries (its running minimum is never lowered), so the
        // result is that of the last ray.
        const int centre = top_share > 1.0 - power_percent ? strongest : P - 1;
        for (int k = lane; k < P; k += 64) d[k] = sphere_distance(xyz + 3 * k, xyz + 3 * centre);
        __builtin_amdgcn_wave_barrier();
        // nearest first until the share is gathered (:310-326)
        int n_in = 0;
        double gathered = 0;
        for (int it = 0; it < P; ++it) {
            if (gathered / total >= power_percent) break;
            double v = INFINITY;
            int vi = -1;
            for (int k = lane; k < P; k += 64)
                if (vi < 0 || d[k] < v) {
                    v = d[k];
                    vi = k;
                }
            wave_arg<false>(v, vi);
            if (lane == 0) {
                ord[n_in] = vi;
                d[vi] = INFINITY;
            }
            gathered += pw[vi];
            ++n_in;
            __builtin_amdgcn_wave_barrier();
        }
        // the two gathered rays furthest apart (:328-342), as positions in the gathered list
        int far_a = -1, far_b = -1;
        if (n_in == 2) {
            far_a = 0;
            far_b = 1;
        } else {
            double furthest = 0;
            for (int i = 0; i < n_in; ++i) {
                const double *pi = xyz + 3 * ord[i];
                double v = -INFINITY;
                int vi = -1;
                for (int k = lane; k < n_in; k += 64) {
                    const double dk = sphere_distance(xyz + 3 * ord[k], pi);
                    if (dk > v) {
                        v = dk;
                        vi = k;
                    }
                }
                wave_arg<true>(v, vi);
                if (v > furthest) {
                    furthest = v;
                    far_a = i;
                    far_b = vi;
                }
            }
        }
        if (far_a >= 0) {
            // the reference reads the angles of the rays at those positions of the receiver's own list (:367-368)
            win_el = angle_distance(el[far_a], el[far_b]);
            win_az = angle_distance(az[far_a], az[far_b]);
        } else {
            win_el = win_az = NAN;  // the reference fails here (nothing to unpack)
        }
    }
    if (lane == 0) {
        o[0] = delay;
        o[1] = rms_az;
        o[2] = rms_el;
        o[3] = percent;
        o[4] = has_los ? 1.0 : 0.0;
        o[5] = win_az;
        o[6] = win_el;
        o[7] = (double)P;
    }
}

}  // namespace

extern "C" int rt_channel_stats(int device, int64_t n_groups, const int64_t *group_off, const double *power,
                                const double *time, const double *el_deg, const double *az_deg, const double *xyz,
                                const uint8_t *is_los, int32_t n_strongest, double power_percent, double *out,
                                int on_device) {
    if (n_groups < 0 || !group_off || !out) RT_FAIL("rt_channel_stats: bad argument");
    if (n_strongest < 0 || !(power_percent > 0.0 && power_percent <= 1.0)) RT_FAIL("rt_channel_stats: bad parameter");
    if (n_groups == 0) return 0;
    if (n_groups > 0x7fffffffLL) RT_FAIL("rt_channel_stats: too many receivers");
    RT_HIP(rt_use_device(device));
    int64_t n_paths = 0;
    int64_t *d_off = nullptr;
    double *d_val = nullptr, *d_out = nullptr, *d_dist = nullptr;
    uint8_t *d_los = nullptr;
    int32_t *d_order = nullptr;
    StatsIn in{};
    int rc = -1;
    do {
        if (on_device) {
            int64_t last = 0;
            if (hipMemcpy(&last, group_off + n_groups, sizeof(last), hipMemcpyDeviceToHost) != hipSuccess) {
                rt_set_error("rt_channel_stats: cannot read the offsets");
                break;
            }
            n_paths = last;
            in = StatsIn{group_off, power, time, el_deg, az_deg, xyz, is_los};
            d_out = out;
        } else {
            for (int64_t g = 0; g < n_groups; ++g)
                if (group_off[g + 1] < group_off[g]) {
                    rt_set_error("rt_channel_stats: offsets must not decrease");
                    return -1;
                }
            if (group_off[0] != 0) RT_FAIL("rt_channel_stats: offsets must start at 0");
            n_paths = group_off[n_groups];
        }
        if (n_paths > 0 && (!power || !time || !el_deg || !az_deg || !xyz || !is_los)) {
            rt_set_error("rt_channel_stats: null array");
            break;
        }
        const size_t np = (size_t)(n_paths > 0 ? n_paths : 1);
        if (hipMalloc((void **)&d_dist, sizeof(double) * np) != hipSuccess ||
            hipMalloc((void **)&d_order, sizeof(int32_t) * np) != hipSuccess) {
            rt_set_error("rt_channel_stats: out of device memory");
            break;
        }
        if (!on_device) {
            // one buffer: power | time | el | az | xyz[3]
            if (hipMalloc((void **)&d_off, sizeof(int64_t) * (n_groups + 1)) != hipSuccess ||
                hipMalloc((void **)&d_val, sizeof(double) * 7 * np) != hipSuccess ||
                hipMalloc((void **)&d_los, np) != hipSuccess ||
                hipMalloc((void **)&d_out, sizeof(double) * 8 * n_groups) != hipSuccess) {
                rt_set_error("rt_channel_stats: out of device memory");
                break;
            }
            bool ok = hipMemcpy(d_off, group_off, sizeof(int64_t) * (n_groups + 1), hipMemcpyHostToDevice) == hipSuccess;
            if (n_paths > 0) {
                ok = ok && hipMemcpy(d_val, power, sizeof(double) * n_paths, hipMemcpyHostToDevice) == hipSuccess;
                ok = ok && hipMemcpy(d_val + np, time, sizeof(double) * n_paths, hipMemcpyHostToDevice) == hipSuccess;
                ok = ok && hipMemcpy(d_val + 2 * np, el_deg, sizeof(double) * n_paths, hipMemcpyHostToDevice) == hipSuccess;
                ok = ok && hipMemcpy(d_val + 3 * np, az_deg, sizeof(double) * n_paths, hipMemcpyHostToDevice) == hipSuccess;
                ok = ok && hipMemcpy(d_val + 4 * np, xyz, sizeof(double) * 3 * n_paths, hipMemcpyHostToDevice) == hipSuccess;
                ok = ok && hipMemcpy(d_los, is_los, (size_t)n_paths, hipMemcpyHostToDevice) == hipSuccess;
            }
            if (!ok) {
                rt_set_error("rt_channel_stats: copy to the device failed");
                break;
            }
            in = StatsIn{d_off, d_val, d_val + np, d_val + 2 * np, d_val + 3 * np, d_val + 4 * np, d_los};
        }
        hipLaunchKernelGGL(k_channel_stats, dim3((unsigned)n_groups), dim3(64), 0, 0, in, n_groups, (int)n_strongest,
                           power_percent, d_dist, d_order, d_out);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) {
            rt_set_error(std::string("rt_channel_stats: ") + hipGetErrorString(e));
            break;
        }
        if (!on_device && hipMemcpy(out, d_out, sizeof(double) * 8 * n_groups, hipMemcpyDeviceToHost) != hipSuccess) {
            rt_set_error("rt_channel_stats: copy from the device failed");
            break;
        }
        rc = 0;
    } while (false);
    if (d_dist) (void)hipFree(d_dist);
    if (d_order) (void)hipFree(d_order);
    if (!on_device) {
        if (d_off) (void)hipFree(d_off);
        if (d_val) (void)hipFree(d_val);
        if (d_los) (void)hipFree(d_los);
        if (d_out) (void)hipFree(d_out);
    }
    return rc;
}
