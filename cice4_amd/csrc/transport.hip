// Horizontal transport by incremental remapping on the device.  Behavioural source:
// source/ice_transport_driver.F90 (transport_remap :179, state_to_tracers :847, tracers_to_state :1012) and
// source/ice_transport_remap.F90 (horizontal_remap :328, make_masks :891, construct_fields :1069,
// limited_gradient :1392, departure_points :1565, locate_triangles :1763, triangle_coordinates :3155,
// transport_integrals :3307, update_fields :3642) of the reference, file:line cited at each piece.
// Configuration: the reference's compile-time choices l_fixed_area = F, integral_order = 3 (cubic, 4-point
// quadrature), l_dp_midpt = T (ice_transport_driver.F90:37-38, 56-60); nghost = 1.
//
// Decomposition: the reference walks compressed cell / edge lists per block, category and triangle group; here
// every list is a dense launch with the list's condition as a predicate, and the six triangle groups of an
// edge are handled by the thread that owns the edge, in the reference's order (sums over groups keep their order).
// All device arrays are level-major ((nx_block, ny_block, nblocks) per level), the layout the halo lists address,
// so each multi-level ghost update is ONE call.  Arithmetic is the reference's, operation by operation
// (compiled with -ffp-contract=off): results are bit-identical (tests/test_gpu_transport.py).
#include "transport.h"

#include <cmath>

namespace cice {

using namespace K;

namespace {

constexpr double eps16 = 1.0e-16;
constexpr double c12 = 12.0, p4 = 0.4, p6 = 0.6;
constexpr double p5625m = -9.0 / 16.0, p52083 = 25.0 / 48.0;
// init_remap :266-319: geometric means of a unit square cell
constexpr double xav = c0, yav = c0, xxav = c1 / c12, yyav = c1 / c12, xyav = c0, xxxav = c0, xxyav = c0, xyyav = c0,
                 yyyav = c0;

struct Cell {
  int b, i, j;  // block, 1-based i, j
  size_t q, c;  // offset in the plane, offset in the level (b*np + q)
};

__device__ __forceinline__ bool cell_of(const TransportKernelArgs& a, Cell& k) {
  const size_t np = (size_t)a.nx * a.ny;
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= np) return false;
  k.b = blockIdx.z;
  k.q = q;
  k.c = (size_t)k.b * np + q;
  k.j = (int)(q / a.nx) + 1;
  k.i = (int)(q - (size_t)(k.j - 1) * a.nx) + 1;
  return true;
}

__device__ __forceinline__ bool physical(const TransportKernelArgs& a, const Cell& k) {
  const int32_t* e = a.blk + 4 * k.b;
  return k.i >= e[0] && k.i <= e[1] && k.j >= e[2] && k.j <= e[3];
}

__device__ __forceinline__ size_t lvl(const TransportKernelArgs& a, int level) {
  return (size_t)level * a.nb * a.nx * a.ny;
}
// level of tracer nt (0-based) of category n (1-based) in tm / tmask / tc / tx / ty
__device__ __forceinline__ int tl(const TransportKernelArgs& a, int n, int nt) { return (n - 1) * a.ntrace + nt; }

// ---- state_to_tracers (driver :847-1003) + make_masks (:891-1059) ------------------------------------
__global__ __launch_bounds__(256) void k_tr_tracers(const TransportKernelArgs a) {
  Cell k;
  if (!cell_of(a, k)) return;
  const int n = blockIdx.y;  // 0 = open water
  if (n == 0) {
    const double m = a.aice0[k.c];
    a.mm[k.c] = m;
    a.mmask[k.c] = m > puny ? c1 : c0;
    return;
  }
  const double ai = a.aicen[lvl(a, n - 1) + k.c];
  double t[TR_MAXTRACE];
#pragma unroll
  for (int nt = 0; nt < TR_MAXTRACE; ++nt) t[nt] = c0;
  const bool ice = ai > puny;
  if (ice) {
    const double vi = a.vicen[lvl(a, n - 1) + k.c], vs = a.vsnon[lvl(a, n - 1) + k.c];
    const double w1 = c1 / ai;
    const double worka = c1 / vi;
    t[0] = vi * w1;  // hice
    t[1] = vs * w1;  // hsno
    const double workb = t[1] > puny ? c1 / vs : c0;
    for (int it = 0; it < a.ntrcr; ++it) t[2 + it] = a.trcrn[lvl(a, (n - 1) * NTRCR + it) + k.c];
    const int kt = 2 + a.ntrcr;
#pragma unroll
    for (int l = 0; l < NILYR; ++l) t[kt + l] = a.eicen[lvl(a, (n - 1) * NILYR + l) + k.c] * worka;  // qice
    if (t[1] > puny) {
#pragma unroll
      for (int l = 0; l < NSLYR; ++l)
        t[kt + NILYR + l] = a.esnon[lvl(a, (n - 1) * NSLYR + l) + k.c] * workb + rhos * Lfresh;      // qsno
    }
  }
  a.mm[lvl(a, n) + k.c] = ai;
  a.mmask[lvl(a, n) + k.c] = ice ? c1 : c0;
  for (int nt = 0; nt < a.ntrace; ++nt) {
    a.tm[lvl(a, tl(a, n, nt)) + k.c] = t[nt];
    a.tmask[lvl(a, tl(a, n, nt)) + k.c] = (a.hasdep[nt] && ice && fabs(t[nt]) > puny) ? c1 : c0;
  }
}

// limited_gradient :1392-1556 at one cell (the caller has checked phimask(i,j) > puny and that the cell is physical)
__device__ __forceinline__ void limited_gradient(const TransportKernelArgs& a, const Cell& k, const double* phi,
                                                 const double* msk, double cnx, double cny, double& gx, double& gy) {
  const size_t c = k.c;
  const int nx = a.nx;
  const double p0 = phi[c];
  auto nb = [&](long off) { return msk[c + off] * phi[c + off] + (c1 - msk[c + off]) * p0; };
  const double phi_nw = nb(nx - 1), phi_n = nb(nx), phi_ne = nb(nx + 1), phi_w = nb(-1), phi_e = nb(1),
               phi_sw = nb(-nx - 1), phi_s = nb(-nx), phi_se = nb(-nx + 1);
  // horizontal_remap hands construct_fields UNIT cells (:560-563, 585-605: worka..workd = c1 in the HTN, HTE, dxt,
  // dyt slots): the reconstruction lives in scaled coordinates, a cell is the square [-1/2, 1/2]^2
  constexpr double dxt = c1, dyt = c1, htn = c1, htns = c1, hte = c1, htew = c1;
  const double gxtmp = (phi_e - p0) / (dxt + dxt) + (p0 - phi_w) / (dxt + dxt);
  const double gytmp = (phi_n - p0) / (dyt + dyt) + (p0 - phi_s) / (dyt + dyt);
  double pmn = fmin(fmin(fmin(fmin(fmin(fmin(fmin(fmin(phi_nw, phi_n), phi_ne), phi_w), p0), phi_e), phi_sw), phi_s), phi_se);
  double pmx = fmax(fmax(fmax(fmax(fmax(fmax(fmax(fmax(phi_nw, phi_n), phi_ne), phi_w), p0), phi_e), phi_sw), phi_s), phi_se);
  pmn = pmn - p0;
  pmx = pmx - p0;
  double w1 = (p5 * htn - cnx) * gxtmp + (p5 * hte - cny) * gytmp;
  double w2 = (p5 * htns - cnx) * gxtmp - (p5 * hte + cny) * gytmp;
  const double w3 = -(p5 * htns + cnx) * gxtmp - (p5 * htew + cny) * gytmp;
  const double w4 = (p5 * htew - cny) * gytmp - (p5 * htn + cnx) * gxtmp;
  const double qmn = fmin(fmin(fmin(w1, w2), w3), w4);
  const double qmx = fmax(fmax(fmax(w1, w2), w3), w4);
  w1 = fabs(qmn) > c0 ? fmax(c0, pmn / qmn) : c1;
  w2 = fabs(qmx) > c0 ? fmax(c0, pmx / qmx) : c1;
  w1 = fmin(fmin(c1, w1), w2);
  gx = w1 * gxtmp;
  gy = w1 * gytmp;
}

// ---- construct_fields :1069-1382 for one (cell, category) --------------------------------------------
__global__ __launch_bounds__(256) void k_tr_fields(const TransportKernelArgs a) {
  Cell k;
  if (!cell_of(a, k)) return;
  const int n = blockIdx.y;
  const bool phys = physical(a, k);
  const size_t L = lvl(a, n) + k.c;
  const double m = a.mm[L];
  double mx = c0, my = c0, mc = c0;
  if (phys && a.hm[k.c] > puny) limited_gradient(a, k, a.mm + lvl(a, n), a.hm, xav, yav, mx, my);
  // cells "where ice is present": category 0 over the whole array, the others over the physical cells
  // (make_masks :930-1057)
  const bool listed = m > puny && (n == 0 || phys);
  if (listed) mc = m - xav * mx - yav * my;
  a.mc[L] = mc;
  a.mx[L] = mx;
  a.my[L] = my;
  if (n == 0) return;
  double mxav = c0, myav = c0;
  if (listed) {
    mxav = (mx * xxav + my * xyav + mc * xav) / m;
    myav = (mx * xyav + my * yyav + mc * yav) / m;
  }
  double mtxav[TR_MAXTRACE], mtyav[TR_MAXTRACE];
  const double* mmask = a.mmask + lvl(a, n);
  for (int nt = 0; nt < a.ntrace; ++nt) {
    const size_t T = lvl(a, tl(a, n, nt));
    const double t = a.tm[T + k.c];
    double tc = c0, tx = c0, ty = c0;
    mtxav[nt] = c0;
    mtyav[nt] = c0;
    if (a.type[nt] == 1) {
      if (phys && mmask[k.c] > puny) limited_gradient(a, k, a.tm + T, mmask, mxav, myav, tx, ty);
      if (listed) {
        tc = t - tx * mxav - ty * myav;
        if (a.hasdep[nt] && a.tmask[T + k.c] > puny) {   // centre of area*tracer :1293-1318
          const double w1 = mc * tc;
          const double w2 = mc * tx + mx * tc;
          const double w3 = mc * ty + my * tc;
          const double w4 = mx * tx;
          const double w5 = mx * ty + my * tx;
          const double w6 = my * ty;
          const double w7 = c1 / (m * t);
          mtxav[nt] = (w1 * xav + w2 * xxav + w3 * xyav + w4 * xxxav + w5 * xxyav + w6 * xyyav) * w7;
          mtyav[nt] = (w1 * yav + w2 * xyav + w3 * yyav + w4 * xxyav + w5 * xyyav + w6 * yyyav) * w7;
        }
      }
    } else if (a.type[nt] == 2) {
      const int nt1 = a.dep[nt];
      const double* tmask1 = a.tmask + lvl(a, tl(a, n, nt1));
      if (phys && tmask1[k.c] > puny) limited_gradient(a, k, a.tm + T, tmask1, mtxav[nt1], mtyav[nt1], tx, ty);
      if (listed) tc = t - tx * mtxav[nt1] - ty * mtyav[nt1];
    } else {
      if (listed) tc = t;
    }
    a.tc[T + k.c] = tc;
    a.tx[T + k.c] = tx;
    a.ty[T + k.c] = ty;
  }
}

// ---- departure_points :1565-1753 (l_dp_midpt = T) ----------------------------------------------------
__global__ __launch_bounds__(256) void k_tr_departure(const TransportKernelArgs a) {
  Cell k;
  if (!cell_of(a, k)) return;
  double dpx = c0, dpy = c0;
  if (physical(a, k)) {
    const size_t c = k.c;
    const int nx = a.nx;
    const double u = a.uvel[c], v = a.vvel[c];
    dpx = -a.dt * u;
    dpy = -a.dt * v;
    if (dpx < -a.HTN[c] || dpx > a.HTN[c + 1] || dpy < -a.HTE[c] || dpy > a.HTE[c + nx])
      atomicMin(a.errkey, ((unsigned long long)1 << 60) | ((unsigned long long)k.b << 40) | (unsigned long long)k.q);
    if (u != c0 || v != c0) {
      dpx = dpx / a.dxu[c];
      dpy = dpy / a.dyu[c];
      const double mpx = p5 * dpx, mpy = p5 * dpy;
      long o2;  // offset of cell (i2, j2) relative to (i, j)
      double mpxt, mpyt;
      if (mpx >= c0 && mpy >= c0) { o2 = nx + 1; mpxt = mpx - p5; mpyt = mpy - p5; }
      else if (mpx < c0 && mpy < c0) { o2 = 0; mpxt = mpx + p5; mpyt = mpy + p5; }
      else if (mpx >= c0 && mpy < c0) { o2 = 1; mpxt = mpx - p5; mpyt = mpy + p5; }
      else { o2 = nx; mpxt = mpx + p5; mpyt = mpy - p5; }
      const size_t c2 = c + o2;
      const double ump = a.uvel[c2 - nx - 1] * (mpxt - p5) * (mpyt - p5) - a.uvel[c2 - nx] * (mpxt + p5) * (mpyt - p5) +
                         a.uvel[c2] * (mpxt + p5) * (mpyt + p5) - a.uvel[c2 - 1] * (mpxt - p5) * (mpyt + p5);
      const double vmp = a.vvel[c2 - nx - 1] * (mpxt - p5) * (mpyt - p5) - a.vvel[c2 - nx] * (mpxt + p5) * (mpyt - p5) +
                         a.vvel[c2] * (mpxt + p5) * (mpyt + p5) - a.vvel[c2 - 1] * (mpxt - p5) * (mpyt + p5);
      dpx = -a.dt * ump;
      dpy = -a.dt * vmp;
    }
  }
  a.dpx[k.c] = dpx;
  a.dpy[k.c] = dpy;
}

// ---- locate_triangles :1763-3146 + triangle_coordinates :3155-3297 (cubic) for one edge ------------------
// The geometry of an edge's departure region (Geo) is worked out once; the six triangle groups are then produced ONE AT
// A TIME (tri_group<G>: the reference's case analysis with only group G's assignments kept -- the conditions are a
// handful of comparisons), so that a thread holds one triangle (9 doubles) instead of six while it integrates.
struct Geo {
  double xdl0, ydl0, xdr0, ydr0;   // departure points of the two corners (edge-local coordinates)
  double xdl, ydl, xdr, ydr;       // ... after the redefinition of points that lie in side cells (:2230-2238)
  double xdm, ydm, xil, yil, xir, yir, xic;
  double afl, afr, afc;
};
struct Tri1 {
  double xp[4], yp[4], area;       // vertex 0 = centroid after triangle_coordinates; 1..3 quadrature points
  int di, dj;                      // iflux - i, jflux - j
  bool on;                         // |triarea| >= eps16 * areafac_c
};

// north: dir = 1, east: dir = 0.  Returns false if the edge has no departure region (both corner points at rest).
__device__ __forceinline__ bool edge_geometry(const TransportKernelArgs& a, const Cell& k, int dir, Geo& q) {
  const size_t c = k.c;
  const int nx = a.nx;
  const size_t cl = dir ? c - 1 : c, cr = dir ? c : c - nx;   // left / right corner of the edge (U points)
  if (!(a.dpx[cl] != c0 || a.dpy[cl] != c0 || a.dpx[cr] != c0 || a.dpy[cr] != c0)) return false;
  q.afl = a.dxu[cl] * a.dyu[cl];
  q.afr = a.dxu[cr] * a.dyu[cr];
  q.afc = p5 * (q.afl + q.afr);
  const double dxl = a.dpx[cl] / a.dxu[cl], dyl = a.dpy[cl] / a.dyu[cl];
  const double dxr = a.dpx[cr] / a.dxu[cr], dyr = a.dpy[cr] / a.dyu[cr];
  const double xcl = -p5, ycl = c0, xcr = p5, ycr = c0;
  double xdl, ydl, xdr, ydr;
  if (dir) { xdl = xcl + dxl; ydl = ycl + dyl; xdr = xcr + dxr; ydr = ycr + dyr; }
  else { xdl = xcl - dyl; ydl = ycl + dxl; xdr = xcr - dyr; ydr = ycr + dxr; }   // rotate trajectory by pi/2
  const double xdm = p5 * (xdr + xdl), ydm = p5 * (ydr + ydl);
  q.xil = xcl; q.yil = (xcl * (ydm - ydl) + xdm * ydl - xdl * ydm) / (xdm - xdl);
  q.xir = xcr; q.yir = (xcr * (ydr - ydm) - xdm * ydr + xdr * ydm) / (xdr - xdm);
  const double md = (ydr - ydl) / (xdr - xdl);
  q.xic = fabs(md) > puny ? xdl - ydl / md : c0;
  q.xdm = xdm; q.ydm = ydm;
  q.xdl0 = xdl; q.ydl0 = ydl; q.xdr0 = xdr; q.ydr0 = ydr;
  // redefine departure points that lie in side cells (:2230-2238)
  if (xdl < xcl) { xdl = q.xil; ydl = q.yil; }
  if (xdr > xcr) { xdr = q.xir; ydr = q.yir; }
  q.xdl = xdl; q.ydl = ydl; q.xdr = xdr; q.ydr = ydr;
  return true;
}

template <int G>
__device__ __forceinline__ void tri_group(const Geo& q, int dir, Tri1& t) {
  // shifts of the cells a triangle can lie in (:1888-1939)
  const int tl_i = dir ? -1 : 1, tl_j = 1, bl_i = dir ? -1 : 0, bl_j = dir ? 0 : 1;
  const int tr_i = 1, tr_j = dir ? 1 : -1, br_i = dir ? 1 : 0, br_j = dir ? 0 : -1;
  const int tc_i = dir ? 0 : 1, tc_j = dir ? 1 : 0, bc_i = 0, bc_j = 0;
  const double xcl = -p5, ycl = c0, xcr = p5, ycr = c0;
  const double afl = q.afl, afr = q.afr, afc = q.afc;
  const double xil = q.xil, yil = q.yil, xir = q.xir, yir = q.yir, xic = q.xic, yic = c0, xdm = q.xdm, ydm = q.ydm;
  const double xicl = xic, yicl = yic, xicr = xic, yicr = yic;   // l_fixed_area = F
  double x1 = c0, y1 = c0, x2 = c0, y2 = c0, x3 = c0, y3 = c0, fact = c0;
  t.di = 0; t.dj = 0;
#define SET(ng, X1, Y1, X2, Y2, X3, Y3, DI, DJ, F) \
  do { if ((ng) == G) { x1 = (X1); y1 = (Y1); x2 = (X2); y2 = (Y2); x3 = (X3); y3 = (Y3); t.di = (DI); t.dj = (DJ); fact = (F); } } while (0)
  if (G <= 2) {   // side triangles, with the departure points as they are (:1950-2228)
    const double xdl = q.xdl0, ydl = q.ydl0, xdr = q.xdr0, ydr = q.ydr0;
  // groups are 1-based in the reference: index g-1 here
    if (yil > c0 && xdl < xcl && ydl >= c0) {
      SET(0, xcl, ycl, xil, yil, xdl, ydl, tl_i, tl_j, -afl);
    } else if (yil < c0 && xdl < xcl && ydl < c0) {
      SET(0, xcl, ycl, xdl, ydl, xil, yil, bl_i, bl_j, afl);
    } else if (yil < c0 && xdl < xcl && ydl >= c0) {
      SET(0, xcl, ycl, xdl, ydl, xic, yic, tl_i, tl_j, afl);
      SET(2, xcl, ycl, xic, yic, xil, yil, bl_i, bl_j, afl);
    } else if (yil > c0 && xdl < xcl && ydl < c0) {
      SET(2, xcl, ycl, xil, yil, xic, yic, tl_i, tl_j, -afl);
      SET(0, xcl, ycl, xic, yic, xdl, ydl, bl_i, bl_j, -afl);
    }
    if (yir > c0 && xdr >= xcr && ydr >= c0) {
      SET(1, xcr, ycr, xdr, ydr, xir, yir, tr_i, tr_j, -afr);
    } else if (yir < c0 && xdr >= xcr && ydr < c0) {
      SET(1, xcr, ycr, xir, yir, xdr, ydr, br_i, br_j, afr);
    } else if (yir < c0 && xdr >= xcr && ydr >= c0) {
      SET(1, xcr, ycr, xic, yic, xdr, ydr, tr_i, tr_j, afr);
      SET(2, xcr, ycr, xir, yir, xic, yic, br_i, br_j, afr);
    } else if (yir > c0 && xdr >= xcr && ydr < c0) {
      SET(2, xcr, ycr, xic, yic, xir, yir, tr_i, tr_j, -afr);
      SET(1, xcr, ycr, xdr, ydr, xic, yic, br_i, br_j, -afr);
    }
  }
  if (G >= 3) {   // central triangles, with the redefined departure points (:2343-2983)
    const double xdl = q.xdl, ydl = q.ydl, xdr = q.xdr, ydr = q.ydr;
  // central triangles (:2343-2983)
    if (ydl >= c0 && ydr >= c0 && ydm >= c0) {
      SET(3, xcl, ycl, xcr, ycr, xdl, ydl, tc_i, tc_j, -afc);
      SET(4, xcr, ycr, xdr, ydr, xdl, ydl, tc_i, tc_j, -afc);
      SET(5, xdl, ydl, xdr, ydr, xdm, ydm, tc_i, tc_j, -afc);
    } else if (ydl >= c0 && ydr >= c0 && ydm < c0) {
      SET(3, xcl, ycl, xicl, yicl, xdl, ydl, tc_i, tc_j, -afc);
      SET(4, xcr, ycr, xdr, ydr, xicr, yicr, tc_i, tc_j, -afc);
      SET(5, xicr, yicr, xicl, yicl, xdm, ydm, bc_i, bc_j, afc);
    } else if (ydl < c0 && ydr < c0 && ydm < c0) {
      SET(3, xcl, ycl, xdl, ydl, xcr, ycr, bc_i, bc_j, afc);
      SET(4, xcr, ycr, xdl, ydl, xdr, ydr, bc_i, bc_j, afc);
      SET(5, xdl, ydl, xdm, ydm, xdr, ydr, bc_i, bc_j, afc);
    } else if (ydl < c0 && ydr < c0 && ydm >= c0) {
      SET(3, xcl, ycl, xdl, ydl, xicl, yicl, bc_i, bc_j, afc);
      SET(4, xcr, ycr, xicr, yicr, xdr, ydr, bc_i, bc_j, afc);
      SET(5, xicl, yicl, xicr, yicr, xdm, ydm, tc_i, tc_j, -afc);
    } else if (ydl >= c0 && ydr < c0 && xic >= c0 && ydm >= c0) {
      SET(3, xcl, ycl, xicr, yicr, xdl, ydl, tc_i, tc_j, -afc);
      SET(4, xcr, ycr, xicr, yicr, xdr, ydr, bc_i, bc_j, afr);
      SET(5, xdl, ydl, xicr, yicr, xdm, ydm, tc_i, tc_j, -afc);
    } else if (ydl >= c0 && ydr < c0 && xic >= c0 && ydm < c0) {
      SET(3, xcl, ycl, xicl, yicl, xdl, ydl, tc_i, tc_j, -afc);
      SET(4, xcr, ycr, xicr, yicr, xdr, ydr, bc_i, bc_j, afr);
      SET(5, xicr, yicr, xicl, yicl, xdm, ydm, bc_i, bc_j, afc);
    } else if (ydl >= c0 && ydr < c0 && xic < c0 && ydm < c0) {
      SET(3, xcl, ycl, xicl, yicl, xdl, ydl, tc_i, tc_j, -afl);
      SET(4, xcr, ycr, xicl, yicl, xdr, ydr, bc_i, bc_j, afc);
      SET(5, xdr, ydr, xicl, yicl, xdm, ydm, bc_i, bc_j, afc);
    } else if (ydl >= c0 && ydr < c0 && xic < c0 && ydm >= c0) {
      SET(3, xcl, ycl, xicl, yicl, xdl, ydl, tc_i, tc_j, -afl);
      SET(4, xcr, ycr, xicr, yicr, xdr, ydr, bc_i, bc_j, afc);
      SET(5, xicl, yicl, xicr, yicr, xdm, ydm, tc_i, tc_j, -afc);
    } else if (ydl < c0 && ydr >= c0 && xic < c0 && ydm >= c0) {
      SET(3, xcl, ycl, xdl, ydl, xicl, yicl, bc_i, bc_j, afl);
      SET(4, xcr, ycr, xdr, ydr, xicl, yicl, tc_i, tc_j, -afc);
      SET(5, xicl, yicl, xdr, ydr, xdm, ydm, tc_i, tc_j, -afc);
    } else if (ydl < c0 && ydr >= c0 && xic < c0 && ydm < c0) {
      SET(3, xcl, ycl, xdl, ydl, xicl, yicl, bc_i, bc_j, afl);
      SET(4, xcr, ycr, xdr, ydr, xicr, yicr, tc_i, tc_j, -afc);
      SET(5, xicr, yicr, xicl, yicl, xdm, ydm, bc_i, bc_j, afc);
    } else if (ydl < c0 && ydr >= c0 && xic >= c0 && ydm < c0) {
      SET(3, xcl, ycl, xdl, ydl, xicr, yicr, bc_i, bc_j, afc);
      SET(4, xcr, ycr, xdr, ydr, xicr, yicr, tc_i, tc_j, -afr);
      SET(5, xicr, yicr, xdl, ydl, xdm, ydm, bc_i, bc_j, afc);
    } else if (ydl < c0 && ydr >= c0 && xic >= c0 && ydm >= c0) {
      SET(3, xcl, ycl, xdl, ydl, xicl, yicl, bc_i, bc_j, afc);
      SET(4, xcr, ycr, xdr, ydr, xicr, yicr, tc_i, tc_j, -afr);
      SET(5, xicl, yicl, xicr, yicr, xdm, ydm, tc_i, tc_j, -afc);
    }
  }
#undef SET
  // triangle area :3029-3050, coordinates relative to the cell that contributes :3086-3117, quadrature points
  // :3268-3293 (cubic)
  const double ar = p5 * ((x2 - x1) * (y3 - y1) - (y2 - y1) * (x3 - x1)) * fact;
  t.on = !(fabs(ar) < eps16 * afc);
  t.area = t.on ? ar : c0;
  if (!t.on) return;
  t.xp[1] = x1; t.yp[1] = y1; t.xp[2] = x2; t.yp[2] = y2; t.xp[3] = x3; t.yp[3] = y3;
#pragma unroll
  for (int v = 1; v <= 3; ++v) {
    if (dir) {
      t.xp[v] = t.xp[v] - c1 * t.di;
      t.yp[v] = t.yp[v] + p5 - c1 * t.dj;
    } else {
      const double w1 = t.xp[v];
      t.xp[v] = t.yp[v] + p5 - c1 * t.di;
      t.yp[v] = -w1 - c1 * t.dj;
    }
  }
  t.xp[0] = p333 * (t.xp[1] + t.xp[2] + t.xp[3]);
  t.yp[0] = p333 * (t.yp[1] + t.yp[2] + t.yp[3]);
#pragma unroll
  for (int v = 1; v <= 3; ++v) {
    t.xp[v] = p4 * t.xp[v] + p6 * t.xp[0];
    t.yp[v] = p4 * t.yp[v] + p6 * t.yp[0];
  }
}

// ---- transport_integrals :3307-3632 (cubic) for one (edge, direction, category) ------------------------
// Every loop is unrolled and every array index a compile-time constant (the dependency of a tracer on an earlier one
// is resolved by an unrolled select): the tracer sums stay in registers.  With run-time indices, a non-inlined
// locate_triangles and all six triangles held at once the kernel sat at 256 VGPRs + 120 B of scratch, one wavefront
// per SIMD.  x / y sums of a tracer are kept only for tracers 0 and 1 (hice, hsno): only those have volume-weighted
// dependents (tracer_type 2 depends on ice or snow volume, ice_transport_driver.F90:120-160), and Transport::init
// creates no other dependency.
struct FluxAcc {
  double mflx, mtflx[TR_MAXTRACE];
};

template <int G>
__device__ __forceinline__ void flux_group(const TransportKernelArgs& a, const Cell& k, const Geo& q, int dir, int n,
                                           const double* mc, const double* mx, const double* my, FluxAcc& f) {
  Tri1 t;
  tri_group<G>(q, dir, t);
  if (!t.on) return;
  const size_t c2 = k.c + (long)t.dj * a.nx + t.di;
  const double mc2 = mc[c2], mx2 = mx[c2], my2 = my[c2];
  const double m0 = p5625m * (mc2 + t.xp[0] * mx2 + t.yp[0] * my2);
  const double m1 = p52083 * (mc2 + t.xp[1] * mx2 + t.yp[1] * my2);
  const double m2 = p52083 * (mc2 + t.xp[2] * mx2 + t.yp[2] * my2);
  const double m3 = p52083 * (mc2 + t.xp[3] * mx2 + t.yp[3] * my2);
  const double msum = m0 + m1 + m2 + m3;
  f.mflx = f.mflx + t.area * msum;
  if (n == 0) return;
  double w0 = m0 * t.xp[0], w1 = m1 * t.xp[1], w2 = m2 * t.xp[2], w3 = m3 * t.xp[3];
  const double mxsum = w0 + w1 + w2 + w3;
  const double mxxsum = w0 * t.xp[0] + w1 * t.xp[1] + w2 * t.xp[2] + w3 * t.xp[3];
  const double mxysum = w0 * t.yp[0] + w1 * t.yp[1] + w2 * t.yp[2] + w3 * t.yp[3];
  w0 = m0 * t.yp[0]; w1 = m1 * t.yp[1]; w2 = m2 * t.yp[2]; w3 = m3 * t.yp[3];
  const double mysum = w0 + w1 + w2 + w3;
  const double myysum = w0 * t.yp[0] + w1 * t.yp[1] + w2 * t.yp[2] + w3 * t.yp[3];
  double mtsum[TR_MAXTRACE], mtxsum[2] = {c0, c0}, mtysum[2] = {c0, c0};
#pragma unroll
  for (int nt = 0; nt < TR_MAXTRACE; ++nt) mtsum[nt] = c0;
#pragma unroll
  for (int nt = 0; nt < TR_MAXTRACE; ++nt) {
    if (nt >= a.ntrace) continue;      // (not `break`: the loop has to unroll completely)
    const size_t T = lvl(a, tl(a, n, nt)) + c2;
    const double tc = a.tc[T], tx = a.tx[T], ty = a.ty[T];
    const int ty_ = a.type[nt], nt1 = a.dep[nt];
    if (ty_ == 1) {
      mtsum[nt] = msum * tc + mxsum * tx + mysum * ty;
      f.mtflx[nt] = f.mtflx[nt] + t.area * mtsum[nt];
      if (nt < 2) {
        mtxsum[nt] = mxsum * tc + mxxsum * tx + mxysum * ty;
        mtysum[nt] = mysum * tc + mxysum * tx + myysum * ty;
      }
    } else if (ty_ == 2) {   // depends on tracer 0 or 1
      const double ds = nt1 == 0 ? mtsum[0] : mtsum[1], dx = nt1 == 0 ? mtxsum[0] : mtxsum[1],
                   dy = nt1 == 0 ? mtysum[0] : mtysum[1];
      mtsum[nt] = ds * tc + dx * tx + dy * ty;
      f.mtflx[nt] = f.mtflx[nt] + t.area * mtsum[nt];
    } else {                 // depends on an earlier tracer of any type
      double ds = c0;
#pragma unroll
      for (int m = 0; m < nt; ++m)
        if (m == nt1) ds = mtsum[m];
      mtsum[nt] = ds * tc;
      f.mtflx[nt] = f.mtflx[nt] + t.area * mtsum[nt];
    }
  }
}

// grid: (cells, 2*(ncat+1), nblocks): blockIdx.y = dir * (ncat+1) + n
__global__ __launch_bounds__(256, 3) void k_tr_fluxes(const TransportKernelArgs a) {
  Cell k;
  if (!cell_of(a, k)) return;
  const int dir = blockIdx.y / (NCAT + 1), n = blockIdx.y % (NCAT + 1);
  const int32_t* e = a.blk + 4 * k.b;
  // north edges: i = ilo..ihi, j = jlo-1..jhi; east edges: i = ilo-1..ihi, j = jlo..jhi (:1885-1939)
  const bool edge = dir ? (k.i >= e[0] && k.i <= e[1] && k.j >= e[2] - 1 && k.j <= e[3])
                        : (k.i >= e[0] - 1 && k.i <= e[1] && k.j >= e[2] && k.j <= e[3]);
  const size_t F = (size_t)dir * (NCAT + 1) * a.nb * a.nx * a.ny;                    // mflx of this direction
  const size_t FT = (size_t)dir * NCAT * a.ntrace * a.nb * a.nx * a.ny;             // mtflx of this direction
  FluxAcc f;
  f.mflx = c0;
#pragma unroll
  for (int nt = 0; nt < TR_MAXTRACE; ++nt) f.mtflx[nt] = c0;
  Geo q;
  if (edge && edge_geometry(a, k, dir, q)) {
    const double* mc = a.mc + lvl(a, n);
    const double* mx = a.mx + lvl(a, n);
    const double* my = a.my + lvl(a, n);
    flux_group<0>(a, k, q, dir, n, mc, mx, my, f);   // the six groups in the reference's order (:3395)
    flux_group<1>(a, k, q, dir, n, mc, mx, my, f);
    flux_group<2>(a, k, q, dir, n, mc, mx, my, f);
    flux_group<3>(a, k, q, dir, n, mc, mx, my, f);
    flux_group<4>(a, k, q, dir, n, mc, mx, my, f);
    flux_group<5>(a, k, q, dir, n, mc, mx, my, f);
  }
  a.mflx[F + lvl(a, n) + k.c] = f.mflx;
  if (n >= 1) {
#pragma unroll
    for (int nt = 0; nt < TR_MAXTRACE; ++nt)
      if (nt < a.ntrace) a.mtflx[FT + lvl(a, tl(a, n, nt)) + k.c] = f.mtflx[nt];
  }
}

// ---- update_fields :3642-3868 for one (physical cell, category) ------------------------------------------
__global__ __launch_bounds__(256) void k_tr_update(const TransportKernelArgs a) {
  Cell k;
  if (!cell_of(a, k)) return;
  if (!physical(a, k)) return;
  const int n = blockIdx.y;
  const size_t N = (size_t)a.nb * a.nx * a.ny;
  const size_t L = lvl(a, n) + k.c;
  const double* fe = a.mflx + lvl(a, n);
  const double* fn = a.mflx + (size_t)(NCAT + 1) * N + lvl(a, n);
  const double told = a.tarear[k.c];
  const double mold = a.mm[L];
  double w1 = fe[k.c] - fe[k.c - 1] + fn[k.c] - fn[k.c - a.nx];
  double m = mold - w1 * told;
  if (m < -puny) {   // negative area: the caller aborts (:3722-3744)
    atomicMin(a.errkey, ((unsigned long long)2 << 60) | ((unsigned long long)k.b << 40) | ((unsigned long long)n << 32) |
                            (unsigned long long)k.q);
    return;
  }
  if (m < c0) m = c0;
  a.mm[L] = m;
  if (n == 0) return;
  double tn[TR_MAXTRACE];
  for (int nt = 0; nt < a.ntrace; ++nt) {
    const size_t T = lvl(a, tl(a, n, nt));
    const double t = a.tm[T + k.c];
    double mtold;
    if (a.type[nt] == 1) mtold = mold * t;
    else if (a.type[nt] == 2) mtold = mold * a.tm[lvl(a, tl(a, n, a.dep[nt])) + k.c] * t;
    else {
      const int nt1 = a.dep[nt], nt2 = a.dep[nt1];
      mtold = mold * a.tm[lvl(a, tl(a, n, nt2)) + k.c] * a.tm[lvl(a, tl(a, n, nt1)) + k.c] * t;
    }
    tn[nt] = c0;
    if (m > c0) {
      const double* te = a.mtflx + T;
      const double* tnn = a.mtflx + (size_t)NCAT * a.ntrace * N + T;
      w1 = te[k.c] - te[k.c - 1] + tnn[k.c] - tnn[k.c - a.nx];
      if (a.type[nt] == 1) {
        tn[nt] = (mtold - w1 * told) / m;
      } else if (a.type[nt] == 2) {
        const int nt1 = a.dep[nt];
        if (fabs(tn[nt1]) > c0) tn[nt] = (mtold - w1 * told) / (m * tn[nt1]);
      } else {
        const int nt1 = a.dep[nt], nt2 = a.dep[nt1];
        if (fabs(tn[nt1]) > c0 && fabs(tn[nt2]) > c0) tn[nt] = (mtold - w1 * told) / (m * tn[nt2] * tn[nt1]);
      }
    }
  }
  // old tracer values of this cell are read above by this thread only: safe to overwrite now
  for (int nt = 0; nt < a.ntrace; ++nt) a.tm[lvl(a, tl(a, n, nt)) + k.c] = tn[nt];
}

// ---- tracers_to_state (driver :1012-1137), every cell of the array -----------------------------------
__global__ __launch_bounds__(256) void k_tr_state(const TransportKernelArgs a) {
  Cell k;
  if (!cell_of(a, k)) return;
  const int n = blockIdx.y;
  if (n == 0) {
    a.aice0[k.c] = a.mm[k.c];
    return;
  }
  const double m = a.mm[lvl(a, n) + k.c];
  if (!(m > c0)) return;
  auto t = [&](int nt) { return a.tm[lvl(a, tl(a, n, nt)) + k.c]; };
  const double vi = m * t(0), vs = m * t(1);
  a.aicen[lvl(a, n - 1) + k.c] = m;
  a.vicen[lvl(a, n - 1) + k.c] = vi;
  a.vsnon[lvl(a, n - 1) + k.c] = vs;
  for (int it = 0; it < a.ntrcr; ++it) a.trcrn[lvl(a, (n - 1) * NTRCR + it) + k.c] = t(2 + it);
  const int kt = 2 + a.ntrcr;
#pragma unroll
  for (int l = 0; l < NILYR; ++l) a.eicen[lvl(a, (n - 1) * NILYR + l) + k.c] = vi * t(kt + l);
#pragma unroll
  for (int l = 0; l < NSLYR; ++l) a.esnon[lvl(a, (n - 1) * NSLYR + l) + k.c] = (t(kt + NILYR + l) - rhos * Lfresh) * vs;
}

}  // namespace

void Transport::init(const cice_transport_config& c, const cice_transport_grid& g) {
  CICE_REQUIRE(dom.nblocks() > 0, "cice_transport_init: no local blocks (call cice_domain_create first)");
  CICE_REQUIRE(dom.overlap == 0, "cice_transport_init: wide-halo slab domains are for the EVP bench only");
  CICE_REQUIRE(c.ntrcr >= 1 && c.ntrcr <= NTRCR, "cice_transport_init: ntrcr out of range");
  ntrcr = c.ntrcr;
  ntrace = 2 + ntrcr + NILYR + NSLYR;
  const size_t np = (size_t)dom.nx_block * dom.ny_block;
  n = (size_t)dom.nblocks() * np;
  std::vector<int32_t> hb;
  for (int gid : dom.local) {
    const Block& b = dom.all[gid];
    hb.insert(hb.end(), {b.ilo, b.ihi, b.jlo, b.jhi});
  }
  blk.alloc(hb.size());
  blk.upload(hb.data(), stream);
  struct G { DevBuf<double>* d; const double* h; };
  G gs[] = {{&HTN, g.HTN}, {&HTE, g.HTE}, {&dxt, g.dxt}, {&dyt, g.dyt}, {&dxu, g.dxu}, {&dyu, g.dyu},
            {&tarear, g.tarear}, {&hm, g.hm}};
  for (G& x : gs) {
    CICE_REQUIRE(x.h != nullptr, "cice_transport_init: NULL grid array");
    x.d->alloc(n);
    x.d->upload(x.h, stream);
  }
  CICE_HIP(hipStreamSynchronize(stream));
  a = TransportKernelArgs{};
  a.nx = dom.nx_block; a.ny = dom.ny_block; a.nb = dom.nblocks(); a.ntrace = ntrace; a.ntrcr = ntrcr;
  // init_transport (driver :81-170): hice, hsno independent; area tracers (depend 0) independent, volume / snow
  // tracers hang on hice / hsno; qice on hice, qsno on hsno
  for (int nt = 0; nt < TR_MAXTRACE; ++nt) { a.type[nt] = 1; a.dep[nt] = -1; a.hasdep[nt] = 0; }
  for (int it = 0; it < ntrcr; ++it) {
    const int d = c.trcr_depend[it];
    CICE_REQUIRE(d >= 0 && d <= 2, "cice_transport_init: trcr_depend must be 0, 1 or 2");
    if (d) { a.type[2 + it] = 2; a.dep[2 + it] = d - 1; }
  }
  for (int l = 0; l < NILYR; ++l) { a.type[2 + ntrcr + l] = 2; a.dep[2 + ntrcr + l] = 0; }
  for (int l = 0; l < NSLYR; ++l) { a.type[2 + ntrcr + NILYR + l] = 2; a.dep[2 + ntrcr + NILYR + l] = 1; }
  for (int nt = 0; nt < ntrace; ++nt)
    if (a.dep[nt] >= 0) a.hasdep[a.dep[nt]] = 1;
  a.blk = blk.p;
  a.HTN = HTN.p; a.HTE = HTE.p; a.dxt = dxt.p; a.dyt = dyt.p; a.dxu = dxu.p; a.dyu = dyu.p;
  a.tarear = tarear.p; a.hm = hm.p;
  aice0.alloc(n); aicen.alloc(n * NCAT); trcrn.alloc(n * NCAT * NTRCR); vicen.alloc(n * NCAT); vsnon.alloc(n * NCAT);
  eicen.alloc(n * NCAT * NILYR); esnon.alloc(n * NCAT * NSLYR); uv.alloc(2 * n);
  const size_t lm = NCAT + 1, lt = (size_t)NCAT * ntrace;
  mm.alloc(n * lm); mmask.alloc(n * lm); tm.alloc(n * lt); tmask.alloc(n * lt);
  ctr.alloc(n * (lm + lt));             // mc | tc   (centre, scalar)
  grad.alloc(n * 2 * (lm + lt));        // mx | tx | my | ty   (centre, vector)
  dp.alloc(2 * n);                      // dpx | dpy (NE corner, vector)
  mflx.alloc(n * 2 * lm); mtflx.alloc(n * 2 * lt);
  key.alloc(1);
  a.aice0 = aice0.p; a.aicen = aicen.p; a.trcrn = trcrn.p; a.vicen = vicen.p; a.vsnon = vsnon.p;
  a.eicen = eicen.p; a.esnon = esnon.p; a.uvel = uv.p; a.vvel = uv.p + n;
  a.mm = mm.p; a.tm = tm.p; a.mmask = mmask.p; a.tmask = tmask.p;
  a.mc = ctr.p; a.tc = ctr.p + n * lm;
  a.mx = grad.p; a.tx = grad.p + n * lm; a.my = grad.p + n * (lm + lt); a.ty = grad.p + n * (2 * lm + lt);
  a.dpx = dp.p; a.dpy = dp.p + n;
  a.mflx = mflx.p; a.mtflx = mtflx.p;
  a.errkey = key.p;
}

// host arrays are (nx,ny,levels,nblocks); the device keeps (nx,ny,nblocks) per level
void Transport::up(double* d, const double* h, int levels) {
  const size_t np = (size_t)dom.nx_block * dom.ny_block;
  const int nb = dom.nblocks();
  if (levels == 1 || nb == 1) {
    CICE_HIP(hipMemcpyAsync(d, h, (size_t)levels * n * 8, hipMemcpyHostToDevice, fan.next()));
    return;
  }
  for (int b = 0; b < nb; ++b)
    CICE_HIP(hipMemcpy2DAsync(d + (size_t)b * np, n * 8, h + (size_t)b * levels * np, np * 8, np * 8, levels,
                              hipMemcpyHostToDevice, fan.next()));
}

// evp -> transport chain: the part of the state the dynamics do not hold starts its way to the device now, on the copy
// streams, and nobody waits for it here -- the next fork() / join() pair on the context's CopyFan (the download of evp,
// at the latest the one in remap()) orders the main stream behind these copies.
void Transport::prefetch(const cice_transport_fields& f) {
  CICE_REQUIRE(n > 0, "cice_transport_init has not been called");
  fan.fork(stream);
  up(aice0.p, f.aice0, 1); up(trcrn.p, f.trcrn, NCAT * NTRCR); up(vsnon.p, f.vsnon, NCAT);
  up(eicen.p, f.eicen, NCAT * NILYR); up(esnon.p, f.esnon, NCAT * NSLYR);
  fan.detach();
}

void Transport::adopt(const double* d_uv, const double* d_aicen, const double* d_vicen) {
  const size_t np = (size_t)dom.nx_block * dom.ny_block;
  const int nb = dom.nblocks();
  CICE_HIP(hipMemcpyAsync(uv.p, d_uv, 2 * n * 8, hipMemcpyDeviceToDevice, stream));
  const double* src[2] = {d_aicen, d_vicen};
  double* dst[2] = {aicen.p, vicen.p};
  for (int k = 0; k < 2; ++k) {
    if (nb == 1) {
      CICE_HIP(hipMemcpyAsync(dst[k], src[k], (size_t)NCAT * n * 8, hipMemcpyDeviceToDevice, stream));
    } else {   // the dynamics keep the host layout (nx,ny,ncat,nblocks)
      for (int b = 0; b < nb; ++b)
        CICE_HIP(hipMemcpy2DAsync(dst[k] + (size_t)b * np, n * 8, src[k] + (size_t)b * NCAT * np, np * 8, np * 8, NCAT,
                                  hipMemcpyDeviceToDevice, stream));
    }
  }
  chained = true;
}

void Transport::remap(double dt, const cice_transport_fields& f, int32_t* l_stop, int32_t* istop, int32_t* jstop) {
  const bool was_chained = chained;   // (read and reset before anything can throw: a failed call must not leave the NEXT one chained)
  chained = false;
  CICE_REQUIRE(n > 0, "cice_transport_init has not been called");
  CICE_REQUIRE(f.aice0 && f.aicen && f.trcrn && f.vicen && f.vsnon && f.eicen && f.esnon && f.uvel && f.vvel,
               "cice_transport_remap: NULL field");
  const size_t np = (size_t)dom.nx_block * dom.ny_block;
  const int nb = dom.nblocks();
  auto down = [&](double* h, const double* d, int levels) {
    if (levels == 1 || nb == 1) {
      CICE_HIP(hipMemcpyAsync(h, d, (size_t)levels * n * 8, hipMemcpyDeviceToHost, fan.next()));
      return;
    }
    for (int b = 0; b < nb; ++b)
      CICE_HIP(hipMemcpy2DAsync(h + (size_t)b * levels * np, np * 8, d + (size_t)b * np, n * 8, np * 8, levels,
                                hipMemcpyDeviceToHost, fan.next()));
  };
  fan.fork(stream);
  if (!was_chained) {   // (chained: five arrays were prefetched during evp, four came from the dynamics' device buffers)
    up(aice0.p, f.aice0, 1); up(aicen.p, f.aicen, NCAT); up(trcrn.p, f.trcrn, NCAT * NTRCR); up(vicen.p, f.vicen, NCAT);
    up(vsnon.p, f.vsnon, NCAT); up(eicen.p, f.eicen, NCAT * NILYR); up(esnon.p, f.esnon, NCAT * NSLYR);
    up(uv.p, f.uvel, 1); up(uv.p + n, f.vvel, 1);
  }
  fan.join();   // (also behind prefetched copies still in flight on the side streams)
  CICE_HIP(hipMemsetAsync(key.p, 0xff, 8, stream));
  a.dt = dt;
  const unsigned gx = (unsigned)((np + 255) / 256);
  const dim3 blk256(256), gcat(gx, NCAT + 1, nb), g1(gx, 1, nb), gflux(gx, 2 * (NCAT + 1), nb);
  const int lm = NCAT + 1, lt = NCAT * ntrace;
  auto stop = [&](int stage) {
    if (stop_stage != stage) return false;
    CICE_HIP(hipStreamSynchronize(stream));
    if (l_stop) *l_stop = 0;
    return true;
  };
  hipLaunchKernelGGL(k_tr_tracers, gcat, blk256, 0, stream, a);
  if (stop(1)) return;
  hipLaunchKernelGGL(k_tr_fields, gcat, blk256, 0, stream, a);
  hipLaunchKernelGGL(k_tr_departure, g1, blk256, 0, stream, a);
  // ghost cells of the departure points and of the reconstructed fields (horizontal_remap :627-648)
  halo.update_r8(dp.p, 2, n, true, LOC_NECORNER, KIND_VECTOR);
  halo.update_r8(ctr.p, lm + lt, n, true, LOC_CENTER, KIND_SCALAR);
  halo.update_r8(grad.p, 2 * (lm + lt), n, true, LOC_CENTER, KIND_VECTOR);
  if (stop(2)) return;
  hipLaunchKernelGGL(k_tr_fluxes, gflux, blk256, 0, stream, a);
  if (stop(3)) return;
  hipLaunchKernelGGL(k_tr_update, gcat, blk256, 0, stream, a);
  if (stop(4)) return;
  hipLaunchKernelGGL(k_tr_state, gcat, blk256, 0, stream, a);
  // bound_state (source/ice_state.F90:162-217) on the device: 65 levels, centre, scalar
  halo.update_r8(aicen.p, NCAT, n, true, LOC_CENTER, KIND_SCALAR);
  for (int c = 0; c < NCAT; ++c)   // trcrn(:,:,1:ntrcr,:,:) only (:206)
    halo.update_r8(trcrn.p + (size_t)c * NTRCR * n, ntrcr, n, true, LOC_CENTER, KIND_SCALAR);
  halo.update_r8(vicen.p, NCAT, n, true, LOC_CENTER, KIND_SCALAR);
  halo.update_r8(vsnon.p, NCAT, n, true, LOC_CENTER, KIND_SCALAR);
  halo.update_r8(eicen.p, NCAT * NILYR, n, true, LOC_CENTER, KIND_SCALAR);
  halo.update_r8(esnon.p, NCAT * NSLYR, n, true, LOC_CENTER, KIND_SCALAR);
  CICE_HIP(hipGetLastError());
  fan.fork(stream);
  down(f.aice0, aice0.p, 1); down(f.aicen, aicen.p, NCAT); down(f.trcrn, trcrn.p, NCAT * NTRCR);
  down(f.vicen, vicen.p, NCAT); down(f.vsnon, vsnon.p, NCAT); down(f.eicen, eicen.p, NCAT * NILYR);
  down(f.esnon, esnon.p, NCAT * NSLYR);
  fan.join();
  unsigned long long hk = 0;
  CICE_HIP(hipMemcpyAsync(&hk, key.p, 8, hipMemcpyDeviceToHost, stream));
  CICE_HIP(hipStreamSynchronize(stream));
  if (l_stop) *l_stop = 0;
  if (istop) *istop = 0;
  if (jstop) *jstop = 0;
  if (hk != ~0ull) {   // 1: departure points out of bounds, 2: negative area
    const size_t q = (size_t)(hk & 0xffffffffull);
    if (l_stop) *l_stop = (int32_t)(hk >> 60);
    if (jstop) *jstop = (int32_t)(q / dom.nx_block) + 1;
    if (istop) *istop = (int32_t)(q % dom.nx_block) + 1;
  }
}

size_t Transport::debug_fetch(int which, double* out) {
  const DevBuf<double>* bufs[9] = {&mm, &tm, &ctr, &grad, &dp, &mflx, &mtflx, &mmask, &tmask};
  CICE_REQUIRE(which >= 0 && which < 9, "bad array id");
  const DevBuf<double>& b = *bufs[which];
  if (out) {
    b.download(out, stream);
    CICE_HIP(hipStreamSynchronize(stream));
  }
  return b.n;
}

// ====================================================================================================================
// advection = 'upwind'
// ====================================================================================================================
namespace {

struct UCell {
  int b, i, j;
  size_t q, c;
};
__device__ __forceinline__ bool ucell_of(const UpwindArgs& a, UCell& k) {
  const size_t np = (size_t)a.nx * a.ny;
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= np) return false;
  k.b = blockIdx.z;
  k.q = q;
  k.c = (size_t)k.b * np + q;
  k.j = (int)(q / a.nx) + 1;
  k.i = (int)(q - (size_t)(k.j - 1) * a.nx) + 1;
  return true;
}
__device__ __forceinline__ bool uphysical(const UpwindArgs& a, const UCell& k) {
  const int32_t* e = a.blk + 4 * k.b;
  return k.i >= e[0] && k.i <= e[1] && k.j >= e[2] && k.j <= e[3];
}
__device__ __forceinline__ size_t ulvl(const UpwindArgs& a, int level) { return (size_t)level * a.nb * a.nx * a.ny; }

// corner velocities averaged to the east / north edges (:718-731); elsewhere the halo update fills in
__global__ __launch_bounds__(256) void k_up_edges(const UpwindArgs a) {
  UCell k;
  if (!ucell_of(a, k)) return;
  double ue = c0, vn = c0;
  if (uphysical(a, k)) {
    ue = p5 * (a.uvel[k.c] + a.uvel[k.c - a.nx]);
    vn = p5 * (a.vvel[k.c] + a.vvel[k.c - 1]);
  }
  a.uee[k.c] = ue;
  a.vnn[k.c] = vn;
}

// state_to_work (:1570-1677) for level blockIdx.y < narr, the enthalpies as they are behind it
__global__ __launch_bounds__(256) void k_up_pack(const UpwindArgs a) {
  UCell k;
  if (!ucell_of(a, k)) return;
  const int L = blockIdx.y;
  double v;
  if (L == 0) {
    v = a.aice0[k.c];
  } else if (L < a.narr) {
    const int per = 3 + a.ntrcr, n = (L - 1) / per, r = (L - 1) - n * per;
    const size_t cn = ulvl(a, n) + k.c;
    if (r == 0) v = a.aicen[cn];
    else if (r == 1) v = a.vicen[cn];
    else if (r == 2) v = a.vsnon[cn];
    else {
      const int it = r - 3;
      const double t = a.trcrn[ulvl(a, n * NTRCR + it) + k.c];
      const int d = a.dep[it];
      v = (d == 0 ? a.aicen[cn] : d == 1 ? a.vicen[cn] : a.vsnon[cn]) * t;
    }
  } else if (L < a.narr + NCAT * NILYR) {
    v = a.eicen[ulvl(a, L - a.narr) + k.c];
  } else {
    v = a.esnon[ulvl(a, L - a.narr - NCAT * NILYR) + k.c];
  }
  a.phi[ulvl(a, L) + k.c] = v;
}

// upwind_field (:1796-1878): the donor-cell fluxes through the four edges of a physical cell, evaluated from the
// field before the update (the reference fills worka / workb for the whole block first)
__device__ __forceinline__ double upw(double dt, double y1, double y2, double av, double h) {
  return p5 * dt * h * ((av + fabs(av)) * y1 + (av - fabs(av)) * y2);   // :1850
}
__global__ __launch_bounds__(256) void k_up_advect(const UpwindArgs a) {
  UCell k;
  if (!ucell_of(a, k)) return;
  const size_t o = ulvl(a, blockIdx.y) + k.c;
  const double* f = a.phi + o;
  double v = f[0];
  if (uphysical(a, k)) {
    const size_t c = k.c;
    const int nx = a.nx;
    const double wa = upw(a.dt, f[0], f[1], a.uee[c], a.HTE[c]);
    const double waw = upw(a.dt, f[-1], f[0], a.uee[c - 1], a.HTE[c - 1]);
    const double wb = upw(a.dt, f[0], f[nx], a.vnn[c], a.HTN[c]);
    const double wbs = upw(a.dt, f[-nx], f[0], a.vnn[c - nx], a.HTN[c - nx]);
    v = v - (wa - waw + wb - wbs) / a.tarea[c];   // :1866-1868
  }
  a.phi2[o] = v;
}

// work_to_state (:1686-1787) with compute_tracers (ice_itd.F90:1482-1590) on every cell of the block
__global__ __launch_bounds__(256) void k_up_unpack(const UpwindArgs a) {
  UCell k;
  if (!ucell_of(a, k)) return;
  const int n = blockIdx.y;   // 0: aice0, 1..ncat
  if (n == 0) {
    a.aice0[k.c] = a.phi2[k.c];
    return;
  }
  const int per = 3 + a.ntrcr, L0 = 1 + (n - 1) * per;
  const size_t cn = ulvl(a, n - 1) + k.c;
  const double ai = a.phi2[ulvl(a, L0) + k.c], vi = a.phi2[ulvl(a, L0 + 1) + k.c], vs = a.phi2[ulvl(a, L0 + 2) + k.c];
  a.aicen[cn] = ai;
  a.vicen[cn] = vi;
  a.vsnon[cn] = vs;
  for (int it = 0; it < a.ntrcr; ++it) {
    const double at = a.phi2[ulvl(a, L0 + 3 + it) + k.c];
    double t;
    if (it == a.it_Tsfc) t = ai > puny ? at / ai : Tocnfrz;
    else if (a.dep[it] == 0) t = ai > puny ? at / ai : c0;
    else if (a.dep[it] == 1) t = vi > c0 ? at / vi : c0;
    else t = vs > c0 ? at / vs : c0;
    a.trcrn[ulvl(a, (n - 1) * NTRCR + it) + k.c] = t;
  }
#pragma unroll
  for (int l = 0; l < NILYR; ++l) {
    const int e = (n - 1) * NILYR + l;
    a.eicen[ulvl(a, e) + k.c] = a.phi2[ulvl(a, a.narr + e) + k.c];
  }
#pragma unroll
  for (int l = 0; l < NSLYR; ++l) {
    const int e = (n - 1) * NSLYR + l;
    a.esnon[ulvl(a, e) + k.c] = a.phi2[ulvl(a, a.narr + NCAT * NILYR + e) + k.c];
  }
}

}  // namespace

void Upwind::init(const cice_transport_config& c, int nt_Tsfc, const double* hHTE, const double* hHTN, const double* htarea) {
  CICE_REQUIRE(dom.nblocks() > 0, "cice_transport_upwind_init: no local blocks (call cice_domain_create first)");
  CICE_REQUIRE(dom.overlap == 0, "cice_transport_upwind_init: wide-halo slab domains are for the EVP bench only");
  CICE_REQUIRE(c.ntrcr >= 1 && c.ntrcr <= NTRCR, "cice_transport_upwind_init: ntrcr out of range");
  CICE_REQUIRE(nt_Tsfc >= 1 && nt_Tsfc <= c.ntrcr, "cice_transport_upwind_init: nt_Tsfc out of range");
  CICE_REQUIRE(hHTE && hHTN && htarea, "cice_transport_upwind_init: NULL grid array");
  const size_t np = (size_t)dom.nx_block * dom.ny_block;
  n = (size_t)dom.nblocks() * np;
  std::vector<int32_t> hb;
  for (int gid : dom.local) {
    const Block& b = dom.all[gid];
    hb.insert(hb.end(), {b.ilo, b.ihi, b.jlo, b.jhi});
  }
  blk.alloc(hb.size());
  blk.upload(hb.data(), stream);
  HTE.alloc(n); HTE.upload(hHTE, stream);
  HTN.alloc(n); HTN.upload(hHTN, stream);
  tarea.alloc(n); tarea.upload(htarea, stream);
  CICE_HIP(hipStreamSynchronize(stream));
  a = UpwindArgs{};
  a.nx = dom.nx_block; a.ny = dom.ny_block; a.nb = dom.nblocks(); a.ntrcr = c.ntrcr;
  a.narr = 1 + NCAT * (3 + c.ntrcr);
  a.nlev = a.narr + NCAT * (NILYR + NSLYR);
  a.it_Tsfc = nt_Tsfc - 1;
  for (int it = 0; it < NTRCR; ++it) a.dep[it] = 0;
  for (int it = 0; it < c.ntrcr; ++it) {
    CICE_REQUIRE(c.trcr_depend[it] >= 0 && c.trcr_depend[it] <= 2, "cice_transport_upwind_init: trcr_depend must be 0, 1 or 2");
    a.dep[it] = c.trcr_depend[it];
  }
  uv.alloc(2 * n); edge.alloc(2 * n);
  aice0.alloc(n); aicen.alloc(n * NCAT); trcrn.alloc(n * NCAT * NTRCR); vicen.alloc(n * NCAT); vsnon.alloc(n * NCAT);
  eicen.alloc(n * NCAT * NILYR); esnon.alloc(n * NCAT * NSLYR);
  phi.alloc(n * a.nlev); phi2.alloc(n * a.nlev);
  a.blk = blk.p; a.HTE = HTE.p; a.HTN = HTN.p; a.tarea = tarea.p;
  a.uvel = uv.p; a.vvel = uv.p + n; a.uee = edge.p; a.vnn = edge.p + n;
  a.aice0 = aice0.p; a.aicen = aicen.p; a.trcrn = trcrn.p; a.vicen = vicen.p; a.vsnon = vsnon.p;
  a.eicen = eicen.p; a.esnon = esnon.p; a.phi = phi.p; a.phi2 = phi2.p;
}

void Upwind::step(double dt, const cice_transport_fields& f) {
  CICE_REQUIRE(n > 0, "cice_transport_upwind_init has not been called");
  CICE_REQUIRE(f.aice0 && f.aicen && f.trcrn && f.vicen && f.vsnon && f.eicen && f.esnon && f.uvel && f.vvel,
               "cice_transport_upwind: NULL field");
  const size_t np = (size_t)dom.nx_block * dom.ny_block;
  const int nb = dom.nblocks();
  // host arrays are (nx,ny,levels,nblocks); the device keeps (nx,ny,nblocks) per level
  auto copy = [&](double* d, double* h, int levels, bool up) {
    if (levels == 1 || nb == 1) {
      if (up) CICE_HIP(hipMemcpyAsync(d, h, (size_t)levels * n * 8, hipMemcpyHostToDevice, fan.next()));
      else CICE_HIP(hipMemcpyAsync(h, d, (size_t)levels * n * 8, hipMemcpyDeviceToHost, fan.next()));
      return;
    }
    for (int b = 0; b < nb; ++b) {
      if (up) CICE_HIP(hipMemcpy2DAsync(d + (size_t)b * np, n * 8, h + (size_t)b * levels * np, np * 8, np * 8, levels,
                                        hipMemcpyHostToDevice, fan.next()));
      else CICE_HIP(hipMemcpy2DAsync(h + (size_t)b * levels * np, np * 8, d + (size_t)b * np, n * 8, np * 8, levels,
                                     hipMemcpyDeviceToHost, fan.next()));
    }
  };
  struct S { DevBuf<double>* d; double* h; int levels; };
  S st[] = {{&aice0, f.aice0, 1}, {&aicen, f.aicen, NCAT}, {&trcrn, f.trcrn, NCAT * NTRCR}, {&vicen, f.vicen, NCAT},
            {&vsnon, f.vsnon, NCAT}, {&eicen, f.eicen, NCAT * NILYR}, {&esnon, f.esnon, NCAT * NSLYR}};
  fan.fork(stream);
  for (S& x : st) copy(x.d->p, x.h, x.levels, true);
  copy(uv.p, const_cast<double*>(f.uvel), 1, true);
  copy(uv.p + n, const_cast<double*>(f.vvel), 1, true);
  fan.join();
  a.dt = dt;
  const unsigned gx = (unsigned)((np + 255) / 256);
  const dim3 blk256(256);
  hipLaunchKernelGGL(k_up_edges, dim3(gx, 1, nb), blk256, 0, stream, a);
  halo.update_r8(a.uee, 1, n, true, LOC_EFACE, KIND_VECTOR);   // :735-738
  halo.update_r8(a.vnn, 1, n, true, LOC_NFACE, KIND_VECTOR);
  hipLaunchKernelGGL(k_up_pack, dim3(gx, a.nlev, nb), blk256, 0, stream, a);
  hipLaunchKernelGGL(k_up_advect, dim3(gx, a.nlev, nb), blk256, 0, stream, a);
  hipLaunchKernelGGL(k_up_unpack, dim3(gx, NCAT + 1, nb), blk256, 0, stream, a);
  // bound_state (source/ice_state.F90:162-217), :826-829
  halo.update_r8(aicen.p, NCAT, n, true, LOC_CENTER, KIND_SCALAR);
  for (int c = 0; c < NCAT; ++c)
    halo.update_r8(trcrn.p + (size_t)c * NTRCR * n, a.ntrcr, n, true, LOC_CENTER, KIND_SCALAR);
  halo.update_r8(vicen.p, NCAT, n, true, LOC_CENTER, KIND_SCALAR);
  halo.update_r8(vsnon.p, NCAT, n, true, LOC_CENTER, KIND_SCALAR);
  halo.update_r8(eicen.p, NCAT * NILYR, n, true, LOC_CENTER, KIND_SCALAR);
  halo.update_r8(esnon.p, NCAT * NSLYR, n, true, LOC_CENTER, KIND_SCALAR);
  CICE_HIP(hipGetLastError());
  fan.fork(stream);
  for (S& x : st) copy(x.d->p, x.h, x.levels, false);
  fan.join();
  CICE_HIP(hipStreamSynchronize(stream));
}

}  // namespace cice
