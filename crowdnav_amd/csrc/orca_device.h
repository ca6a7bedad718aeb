// ORCA (optimal reciprocal collision avoidance) agent-agent solve, one wave lane per (env, agent).
//
// Replaces what `sim.doStep(); sim.getAgentVelocity(0)` does for the calling agent in
//   /root/reference crowd_sim/envs/policy/orca.py:128-129
// i.e. the un-vendored RVO2 library's neighbour selection, half-plane construction and the three
// incremental linear programs.  All arithmetic is IEEE binary32 with NO
// contraction (the translation unit is built with -ffp-contract=off), correctly rounded / and sqrt
// (hipcc default), vector / scalar = multiply by the reciprocal.
//
// Data placement: the per-lane work lists (<= 10 neighbours, <= 10 half-planes, <= 9 projected
// half-planes) are dynamically indexed, so they live in LDS as [slot][lane] planes: lane l touches word
// slot*64 + l, which is conflict-free for every ds_read/ds_write_b32 of a wave (32 lanes of a group hit
// 32 distinct banks).  Everything else stays in VGPRs.
#pragma once
#include <hip/hip_runtime.h>

namespace cn {

constexpr int kWave = 64;      // gfx950 wavefront width; one workgroup = one wave here
constexpr int kMaxNb = 10;     // RVO2 maxNeighbors supported (the reference hard-codes 10, orca.py:62)
constexpr float kRvoEps = 0.00001f;

// A lane's view of one LDS work list of half-planes: component planes of kMaxNb slots x 64 lanes.
struct Planes {
    float* px;  // point.x   of slot k at px[k * kWave]
    float* py;
    float* dx;  // direction.x
    float* dy;
};

struct OrcaLds {
    float pt[2][4][kMaxNb][kWave];  // [0] = ORCA half-planes, [1] = projected half-planes (3-D LP fallback)
    float nb_dist[kMaxNb][kWave];   // neighbour list: squared distance, ascending
    int nb_lane[kMaxNb][kWave];     //                 LDS lane of that neighbour
};

__device__ __forceinline__ Planes planes_of(OrcaLds& s, int which, int lane) {
    Planes p;
    p.px = &s.pt[which][0][0][lane];
    p.py = &s.pt[which][1][0][lane];
    p.dx = &s.pt[which][2][0][lane];
    p.dy = &s.pt[which][3][0][lane];
    return p;
}

// 1-D program on half-plane `k`: optimise along its boundary inside the speed disc and the earlier
// half-planes (Appendix A.5).
__device__ inline bool lp_on_line(const Planes& L, int k, float radius, float ox, float oy, bool dir_opt,
                                  float& rx, float& ry) {
    const float px = L.px[k * kWave], py = L.py[k * kWave];
    const float dx = L.dx[k * kWave], dy = L.dy[k * kWave];
    const float dp = px * dx + py * dy;
    const float disc = (dp * dp + radius * radius) - (px * px + py * py);
    if (disc < 0.0f) return false;
    const float root = sqrtf(disc);
    float t_lo = -dp - root;
    float t_hi = -dp + root;
    for (int i = 0; i < k; ++i) {
        const float qx = L.px[i * kWave], qy = L.py[i * kWave];
        const float ex = L.dx[i * kWave], ey = L.dy[i * kWave];
        const float den = dx * ey - dy * ex;
        const float num = ex * (py - qy) - ey * (px - qx);
        if (fabsf(den) <= kRvoEps) {
            if (num < 0.0f) return false;
            continue;
        }
        const float t = num / den;
        if (den >= 0.0f) {
            t_hi = (t < t_hi) ? t : t_hi;
        } else {
            t_lo = (t_lo < t) ? t : t_lo;
        }
        if (t_lo > t_hi) return false;
    }
    float t;
    if (dir_opt) {
        t = (ox * dx + oy * dy > 0.0f) ? t_hi : t_lo;
    } else {
        t = dx * (ox - px) + dy * (oy - py);
        if (t < t_lo) {
            t = t_lo;
        } else if (t > t_hi) {
            t = t_hi;
        }
    }
    rx = px + t * dx;
    ry = py + t * dy;
    return true;
}

// 2-D program over n half-planes (Appendix A.4); returns the index of the first infeasible one, or n.
__device__ inline int lp_planar(const Planes& L, int n, float radius, float ox, float oy, bool dir_opt,
                                float& rx, float& ry) {
    if (dir_opt) {
        rx = ox * radius;
        ry = oy * radius;
    } else if (ox * ox + oy * oy > radius * radius) {
        const float inv = 1.0f / sqrtf(ox * ox + oy * oy);
        const float ux = ox * inv, uy = oy * inv;
        rx = ux * radius;
        ry = uy * radius;
    } else {
        rx = ox;
        ry = oy;
    }
    for (int i = 0; i < n; ++i) {
        const float ex = L.dx[i * kWave], ey = L.dy[i * kWave];
        if (ex * (L.py[i * kWave] - ry) - ey * (L.px[i * kWave] - rx) > 0.0f) {
            const float kx = rx, ky = ry;
            if (!lp_on_line(L, i, radius, ox, oy, dir_opt, rx, ry)) {
                rx = kx;
                ry = ky;
                return i;
            }
        }
    }
    return n;
}

// Fallback when the planar program is infeasible: minimise the maximum penetration (Appendix A.6).
__device__ inline void lp_relaxed(const Planes& L, const Planes& P, int n, int begin, float radius,
                                  float& rx, float& ry) {
    float distance = 0.0f;
    for (int i = begin; i < n; ++i) {
        const float pix = L.px[i * kWave], piy = L.py[i * kWave];
        const float dix = L.dx[i * kWave], diy = L.dy[i * kWave];
        if (dix * (piy - ry) - diy * (pix - rx) > distance) {
            int m = 0;
            for (int j = 0; j < i; ++j) {
                const float pjx = L.px[j * kWave], pjy = L.py[j * kWave];
                const float djx = L.dx[j * kWave], djy = L.dy[j * kWave];
                const float d = dix * djy - diy * djx;
                float qx, qy;
                if (fabsf(d) <= kRvoEps) {
                    if (dix * djx + diy * djy > 0.0f) continue;  // same direction: j adds nothing
                    qx = 0.5f * (pix + pjx);
                    qy = 0.5f * (piy + pjy);
                } else {
                    const float t = (djx * (piy - pjy) - djy * (pix - pjx)) / d;
                    qx = pix + t * dix;
                    qy = piy + t * diy;
                }
                const float ex = djx - dix, ey = djy - diy;
                const float inv = 1.0f / sqrtf(ex * ex + ey * ey);
                P.px[m * kWave] = qx;
                P.py[m * kWave] = qy;
                P.dx[m * kWave] = ex * inv;
                P.dy[m * kWave] = ey * inv;
                ++m;
            }
            const float kx = rx, ky = ry;
            if (lp_planar(P, m, radius, -diy, dix, true, rx, ry) < m) {
                rx = kx;
                ry = ky;
            }
            distance = dix * (piy - ry) - diy * (pix - rx);
        }
    }
}

struct OrcaParams {
    float neighbor_dist;
    float inv_time_horizon;  // 1.0f / timeHorizon
    float inv_time_step;     // 1.0f / timeStep
    int max_neighbors;
};

// Offer one candidate to the lane's neighbour list (Appendix A.2): keep the <= max_neighbors nearest within
// range, ascending, strict '<' so ties keep visit order and a tie with the current worst is rejected.
__device__ __forceinline__ void offer_neighbor(OrcaLds& s, int lane, int cand_lane, float d2, int max_nb,
                                               int& count, float& range_sq) {
    if (d2 < range_sq) {
        if (count < max_nb) ++count;
        int i = count - 1;
        while (i != 0 && d2 < s.nb_dist[i - 1][lane]) {
            s.nb_dist[i][lane] = s.nb_dist[i - 1][lane];
            s.nb_lane[i][lane] = s.nb_lane[i - 1][lane];
            --i;
        }
        s.nb_dist[i][lane] = d2;
        s.nb_lane[i][lane] = cand_lane;
        if (count == max_nb) range_sq = s.nb_dist[count - 1][lane];
    }
}

// Build one ORCA half-plane (Appendix A.3) for the neighbour whose staged kinematics are (opx,opy,ovx,ovy).
__device__ __forceinline__ void make_half_plane(const OrcaParams& P, float spx, float spy, float svx,
                                                float svy, float opx, float opy, float ovx, float ovy,
                                                float rsum, float& lpx, float& lpy, float& ldx, float& ldy) {
    const float rpx = opx - spx, rpy = opy - spy;
    const float rvx = svx - ovx, rvy = svy - ovy;
    const float dist_sq = rpx * rpx + rpy * rpy;
    const float rsum_sq = rsum * rsum;
    float ux, uy;
    if (dist_sq > rsum_sq) {
        const float wx = rvx - P.inv_time_horizon * rpx;
        const float wy = rvy - P.inv_time_horizon * rpy;
        const float wlen_sq = wx * wx + wy * wy;
        const float dot1 = wx * rpx + wy * rpy;
        if (dot1 < 0.0f && dot1 * dot1 > rsum_sq * wlen_sq) {
            const float wlen = sqrtf(wlen_sq);
            const float inv = 1.0f / wlen;
            const float nx = wx * inv, ny = wy * inv;
            ldx = ny;
            ldy = -nx;
            const float k = rsum * P.inv_time_horizon - wlen;
            ux = k * nx;
            uy = k * ny;
        } else {
            const float leg = sqrtf(dist_sq - rsum_sq);
            const float inv = 1.0f / dist_sq;
            if (rpx * wy - rpy * wx > 0.0f) {
                ldx = (rpx * leg - rpy * rsum) * inv;
                ldy = (rpx * rsum + rpy * leg) * inv;
            } else {
                ldx = -((rpx * leg + rpy * rsum) * inv);
                ldy = -((-rpx * rsum + rpy * leg) * inv);
            }
            const float dot2 = rvx * ldx + rvy * ldy;
            ux = dot2 * ldx - rvx;
            uy = dot2 * ldy - rvy;
        }
    } else {
        const float wx = rvx - P.inv_time_step * rpx;
        const float wy = rvy - P.inv_time_step * rpy;
        const float wlen = sqrtf(wx * wx + wy * wy);
        const float inv = 1.0f / wlen;
        const float nx = wx * inv, ny = wy * inv;
        ldx = ny;
        ldy = -nx;
        const float k = rsum * P.inv_time_step - wlen;
        ux = k * nx;
        uy = k * ny;
    }
    lpx = svx + 0.5f * ux;
    lpy = svy + 0.5f * uy;
}

}  // namespace cn
