// Device-resident EVP dynamics: state, kernels' launch wrappers.
#pragma once
#include <map>
#include <memory>
#include <functional>
#include <vector>

#include "common.h"
#include "domain.h"
#include "halo.h"

namespace cice {

struct SubArgs;  // kernel argument block (evp.hip)
struct SkewArgs; // ... of the sweep kernel

struct EvpScalars {  // set_evp_parameters, ice_dyn_evp.F90:535-577
  double dtei, dte2T, denom1, denom2, rcon, ecci;
  int ndte, evp_damping;
  void set(double dt, int ndte_, int damping);
};

// (evp.hip) one strip's step of the measured balancing of the sweep's segments
double balance_strip(int rows, int n, const int* e, const double* d, const double* w, const unsigned char* act, double* cost,
                     int* ne);

class Evp {
 public:
  Evp(const Domain& d, Halo& h, hipStream_t s, CopyFan& f) : dom(d), halo(h), stream(s), fan(f) {}
  ~Evp();
  void init(const cice_evp_config& cfg, const cice_evp_grid& g);
  void upload(const cice_evp_fields& f);
  void adopt_state(const double* d_aicen, const double* d_vicen, const double* d_vsnon);
  void download(cice_evp_fields& f);
  void prepare(double dt);
  void subcycles(int ksub0, int nsub, float* elapsed_ms);
  void finish();
  void step(double dt) {
    prepare(dt);
    subcycles(1, sc.ndte, nullptr);
    finish();
  }
  // evp(dt) on host arrays (cice_evp): upload, step, download as ONE pipeline -- what prepare() has finished travels to
  // the host while the subcycle loop runs; options "keep_state" / "lazy_stresses" (set_option) leave planes where they are
  void run(double dt, cice_evp_fields& f, const std::function<void()>& while_looping = nullptr);
  void download_stresses(cice_evp_fields& f);   // the 12 stresses of the current state ("lazy_stresses": cice_evp left them)
  void forget_host_state() { io_valid = false; }   // the host copies of the io fields may be newer than the device's
  void set_option(const char* key, int value);
  void active_cells(long long* nt, long long* nu);
  long long debug_read(const char* what, long long* out, long long cap);
  bool derives_metrics() const;
  int tile_waves() const { return waves; }
  int tile_rows() const { return rows_per_wave; }
  bool can_reside() const;   // the whole subcycle loop in one launch, state in registers (k_evp_resident)
  bool can_reside_peer() const;  // the same on one slab of a domain cut across ranks, neighbours' buffers mapped (peer_connect)
  void peer_export(void* out[3]);  // this rank's exchange copies and remote-progress words (device pointers)
  void peer_connect(int side, void* xu0, void* xu1, void* rprog, long long peer_n);
  void peer_connect_rank(int rank, void* xu0, void* xu1, void* rprog, long long peer_n);   // any neighbouring rank (cartesian layouts)
  std::vector<int> peer_ranks() const;   // the ranks this rank's block exchanges ghost cells with, ascending
  int resident_waves() const;  // its wavefronts per workgroup (0: grid too large)
  // device copies the transport may take over right after evp(dt) (cice_transport_chain): u | v of the current state,
  // aicen, vicen as uploaded (host layout)
  const double* d_uv() const { return uv[cur].p; }
  const double* d_aicen() const { return aicen.p; }
  const double* d_vicen() const { return vicen.p; }
  int last_launches = 0;     // subcycle-loop kernel launches of the last subcycles() call (1: the one-launch loop)
  bool peer_buffers_fine() const { return res_xu[0].fine && res_xu[1].fine && res_rprog.fine; }   // what other devices write / poll is fine-grained memory
  bool resident_dense() const; // three 4-wavefront workgroups per CU instead of one workgroup per CU
  bool granules_in_use() const { return can_reside() && !halo.multi_rank() && granules_on() && !resident_dense(); }   // the one-launch loop hands its edge velocities on as data-tagged granules
  bool can_skew() const;     // K subcycles per sweep (k_subcycle_skew) on this domain
  bool can_split() const;    // ... and the sweep in front of a wide-halo refresh as edge + interior launches
  bool skew_rows_on() const { return (skew_gen_pct > 0 || skew_fill_on() || balance_on()) && skew_seg_opt == 0; }   // segments of unequal length (build_skew_rows)
  bool skew_fill_on() const;
  int skew_fill_pct() const;
  bool rowact_on() const;    // workgroups of the sweep shrink to the rows that hold ice
  int resident_map();        // the tile map the one-launch loop last chose (k_res_choose_map), -1: none yet
  bool balance_on() const;   // the sweep's segments follow the measured cost of their rows (one block per rank)
  long long balanced_sweeps() const { return bal_sweeps; }
  bool can_trim() const;     // sweeps on wide-halo slabs over tile lists (extension rows trimmed)
  bool pairs_ok() const;     // the sweep's pair layout of the state applies to this domain
  bool can_skew_fold() const;  // the same on a one-block tripole grid: sweeps + a band of top rows per subcycle
  int skew_levels() const;   // its K
  int skew_seg_rows(int K) const;  // rows a workgroup of the sweep owns
  int skew_strips(int K, int* shift) const;   // column strips of a block
  int skew_subs(int K) const;                 // wavefronts per level (3: one 12-wavefront workgroup per CU)
  int skew_blocks(int K) const;    // workgroups per CU it is built for
  int skew_waves_per_simd(int K) const;
  bool can_fuse() const;     // two subcycles per launch on this domain
  int fused_waves() const;   // wavefronts per workgroup of the fused kernel

  // one-block, host-pointer entries with the reference argument lists
  static void stress_host(hipStream_t s, double dt, int ndte, int damping, int nx, int ny, int ksub,
                          int icellt, const int32_t* ti, const int32_t* tj, const double* uvel,
                          const double* vvel, const double* const grid10[10],
                          const double* strength, double* const sig[12], double* const diag[5],
                          double* str);
  static void stepu_host(hipStream_t s, int nx, int ny, int icellu, const int32_t* ui,
                         const int32_t* uj, const double* const in10[10], const double* str,
                         double* const io6[6]);

 private:
  const Domain& dom;
  Halo& halo;
  hipStream_t stream;
  cice_evp_config cfg{};
  EvpScalars sc{};
  bool ready = false, prepared = false, counted = false;
  bool adopted = false;   // aice, vice, vsno, aice0, aicen, vicen came from adopt_state: the next upload may omit them
  int keep_state = 0;        // option: 1 the caller does not change u, v, the stresses, iceumask on the host between two cice_evp calls; 2: and zeroes the flux fields
  bool lazy_sig = false;     // option: cice_evp does not download the stresses (cice_evp_download_stresses does)
  bool io_valid = false;     // the device copies of the io fields are those the last cice_evp left (and nothing has touched them)
  void upload_some(const cice_evp_fields& f, int skip_io);
  void download_some(cice_evp_fields& f, int part);   // part 1: what prepare() has finished, 2: the rest, 3: both
  int waves = 8, rows_per_wave = 1;  // tile = 64 x (waves*rows_per_wave) T-cells
  bool use_graph = true;
  bool comm_graph = false;   // multi-rank loops: capture the RCCL calls too (opt-in)
  bool fuse_on = true;
  bool skew_on = true;       // K subcycles per sweep where the domain allows and the grid is large enough
  int skew_blocks_opt = 0;
  bool fwd_is_ew_wrap = false;   // the on-rank ghost list is the east-west wrap of full-width blocks and nothing else
  int skew_gen_pct = 10;         // see build_skew_rows (15 until the tiles of a strip were dealt their places by weight: profiles/r05_sweep_places_by_strip.txt)
  int skew_fill = 26;            // see build_skew_rows: longer segments for workgroups on CUs that hold fewer of them
  DevBuf<int32_t> skew_rows;
  int skew_rows_key[6] = {0, 0, 0, 0, 0, 0};
  void build_skew_rows(int K, int tiles_x, int tiles_y, int nblocks, int seg_rows);
  // segments balanced by MEASURED cost (balance_after_sweep): host copy of the table, the static weight of every tile's
  // place on the chip, sweeps still to be measured in this tuning phase, loops since the last one
  std::vector<int32_t> rows_host;
  std::vector<double> rows_w;
  int skew_balance = 1, bal_left = 0, bal_every = 96, bal_since = 0, bal_tiles_x = 0, bal_tiles_y = 0;
  long long bal_sweeps = 0;       // sweeps measured so far (cice_evp_get_info "skew_balanced")
  void balance_after_sweep(hipStream_t s);
  struct BalTile { int strip, first, last; };       // rows relative to jlo; last < first: empty
  std::vector<BalTile> bal_tiles;                   // by place in the launch (SkewArgs::tiles)
  std::vector<double> bal_w;                        // static weight of the place
  DevBuf<int32_t> bal_list;
  int bal_nt = 0, bal_strips = 0, bal_slots = 0, bal_gens = 1, bal_per_xcd = 32, bal_seen = 0, bal_k = 4;
  bool bal_recounted = false;
  std::vector<double> bal_got;                      // weight of the places every strip's tiles hold (deal_places)
  bool places_on() const;
  void deal_places(const std::vector<double>& target);
  void bal_upload(hipStream_t s);
  double place_weight(int tile_lin, int nt, int gens, int per_xcd, bool fill) const;
  DevBuf<int32_t> res_map;        // k_res_choose_map: the tile map of the one-launch loop, chosen once per evp(dt)
  bool res_map_stale = true;
  int res_map_opt = -1;
  DevBuf<unsigned char> rowact;   // k_skew_rowact
  DevBuf<int32_t> run_next, run_end;   // k_skew_runs
  int rowact_strips = 0, rowact_k = 0;
  std::vector<unsigned char> rowact_host;   // (balance_after_sweep)
  bool rowact_host_stale = true;
  bool rowact_opt = true;
  int skew_prio = 1;             // rotate the issue priority among the workgroups sharing a CU
  bool skew_debug = false;
  DevBuf<long long> skew_dbg;
  bool stamps_on = false;        // option "stamps": in-kernel clock stamps (diagnostic build -DCICE4_AMD_STAMPS only)
  DevBuf<long long> stamp_buf;
  size_t stamp_used = 0;
  long long* stamp_buffer(size_t workgroups);
  int skew_stagger_ns = 0;   // start delay per workgroup sharing a CU (k_subcycle_skew), 0 = none
  int skew_k_opt = 0, skew_seg_opt = 0;   // forced K / rows per workgroup (tests, tuning), 0 = auto
  long long skew_min_cells = 600000;      // smaller grids keep k_subcycle2 (or the resident loop); measured: 1000 x 800
                                          // 33.8 us per subcycle against 39.7, 720 x 600 a tie, 500 x 400 16.8 against 12.4
  int waves2 = 0;            // fused kernel: wavefronts per workgroup, 0 = auto
  mutable int waves2_auto = 0;  // the automatic choice, once made
  // resident loop (k_evp_resident): one tile per CU for the whole range of subcycles
  CopyFan& fan;                  // the context's side streams for upload / download (many separate host arrays)
  bool resident_on = true, resident_failed = false;
  int res_w_opt = 0;             // forced wavefronts per workgroup (tests), 0 = auto
  bool res_dense = true;         // allow three 4-wavefront workgroups per CU
  int res_prio = 2;              // issue priority among them: 0 none, 1 by dispatch generation, 2 rotating per subcycle
                                 // (gx1: 183.8 k subcycles/s against 177.4 k with 0 or 1, profiles/r04_resident_prio.txt)
  int res_spin_us = 200000;      // bound of every wait inside the resident kernel
  int res_level = 0;             // 0: dense allowed, 1: one workgroup per CU only (after a dense time-out)
  hipEvent_t res_done_ev = nullptr;   // end of the cross-rank loop, polled (run_resident)
  int res_retry_steps = 64;      // evp(dt) calls after which a time-out is forgiven (a co-tenant may have left), 0 = never
  int res_retry_in = 0;          // calls left until then (0: nothing to forgive, or not forgivable)
  int res_occ[5][2][6] = {};     // workgroups of k_evp_resident<W, DAMP, PEER | FOLD | GRAN> one CU holds (last index: plain, PEER, FOLD, GRAN, FOLD + GRAN, PEER + FOLD), 0 = not asked yet
  int res_gran = 1;              // one-rank domains without a fold: edge velocities travel as data-tagged granules (option "resident_granules": 0 never, 1 by the ice cover, 2 always)
  unsigned* res_why = nullptr;   // page-locked: the eight words read back behind every one-launch loop
  bool res_sparse = false;       // the last step's ice cover left most tiles of the loop empty (run_resident reads k_res_choose_map's count)
  bool granules_on() const;
  DevBuf<int32_t> res_src;       // [cells] the owned U-cell whose velocity a cell holds, -1: nobody's
  DevBuf<double> res_xg;         // [2][cells][4] the granule copies (32 bytes per cell and parity)
  DevBuf<double> res_xgr;        // [4][cells][4] FOLD: granule copies of the raw top-row velocities
  int resident_occupancy(int W, bool damp, bool peer);
  int res_w = 0, res_tiles = 0;  // what res_deps was built for
  unsigned res_epoch = 0;
  DevBuf<int32_t> res_deps;
  DevBuf<unsigned> res_prog;     // [tiles * 32] progress words, then the abort word
  DevBuf<double> res_xu[2];      // exchange copies of (u, v)
  bool skew_fold_on = true;      // one-block tripole grids of sweep size: sweeps + a band of top rows (launch_subcycle_skew_fold)
  void launch_subcycle_skew_fold(int ksub, int K);
  void ensure_band(int K);       // buffers and block table of the band (allocations: outside any capture)
  DevBuf<double> band[2];        // the band's two copies of u, v, 12 sigma (full planes; only the top rows are used)
  DevBuf<int32_t> blk_band;
  int band_k = 0;
  bool res_blocks_on = true;     // one-rank domains of several blocks run the one-launch loop as well
  bool res_fold_on = true;       // one-block tripole domains run the one-launch loop with the fold inside
  void build_resident(int W);
  void build_resident_fold(std::vector<int32_t>& src_of, int tiles_x, int W);   // tripole north boundary inside the loop
  DevBuf<int32_t> res_ftab, res_deps2, res_fslot, res_ffwd;
  hipEvent_t res_t0 = nullptr, res_t1 = nullptr;   // subcycles(..., elapsed_ms): the bracket of the one launch of the loop
  hipEvent_t sub_t0 = nullptr, sub_t1 = nullptr;   // ... of every other form of the range
  bool res_time_it = false, res_timed = false;
  DevBuf<unsigned> res_prog2;
  DevBuf<double> res_xraw[2];
  void build_resident_peer(int W);
  void peer_alloc();
  struct Peer { double* xu[2] = {nullptr, nullptr}; unsigned* rprog = nullptr; unsigned n = 0; } peers[8];   // by place in peer_ranks()
  std::vector<int> peer_back = std::vector<int>(8, 0);   // this rank's place in neighbour s's own list of neighbours
  DevBuf<int32_t> res_rslot, res_rfwd, res_pub;
  DevBuf<unsigned> res_rprog;    // [2 sides][RP_MAX tiles][RES_STRIDE]: progress of the neighbours' tiles, written by them
  bool res_peer_built = false, res_peer_agree = true;
  int res_peer_share = 1;
  bool run_resident(int ksub0, int nsub);
  int loop_launches = 0, graph_launches = 0, stats_left = 3;   // subcycle-loop kernels launched by the current / the captured range
  int flips = 0, graph_flips = 0;  // buffer swaps since the counter was reset / in the captured loop
  bool derive_ok = false, derive_on = true;  // metrics recomputed from HTN/HTE (verified at init)
  size_t n = 0;  // nblocks*ny*nx
  int cur = 0;   // which ping-pong copy of u, v, sigma holds the current values

  // grid
  DevBuf<double> dxt, dyt, dxhy, dyhx, cxp, cyp, cxm, cym, tarea, uarea, tarear, uarear, tinyarea,
      fcor, HTN, HTE;
  DevBuf<int32_t> tmask, umask, blk;  // blk: ilo,ihi,jlo,jhi per block
  DevBuf<double> uarena;   // aiu, uocn, vocn, forcex, forcey, umassdtei, fm, uarear live here (views below)
  DevBuf<double> uar4, hnhe;   // the sweep kernel's interleaved copies: 4 planes of pairs of the above; {HTN, HTE} pairs
  DevBuf<int32_t> skew_msk;    // bit 0: icetmask == 1, bit 1: iceumask != 0
  int skew_subs_opt = 1;       // wavefronts per level of the sweep kernel: 3 = one 12-wavefront workgroup per CU, measured SLOWER
                               // (0.1 degree 286 us per subcycle against 266.5: profiles/r04_sweep_ab_three_wavefronts_per_level.txt)
  mutable int strips_cache[9][2] = {}, strips_cache_shift[9][2] = {};   // [K][S == 3]: strips of a block (0: not computed yet), shift
  bool skew_packed = false;    // ... built for the current prepare()
  void skew_pack();
  // the sweep's pair layout of the state (k_subcycle_skew<.., PAIRS>)
  DevBuf<double> st2[2];
  bool pairs_on = true, in_pairs = false, copies_identical = false;
  void to_pairs();
  void to_planes();
  // in
  DevBuf<double> aice, vice, vsno, aice0, aicen, vicen, strairxT, strairyT, uocn, vocn, ss_tltx,
      ss_tlty;
  // io (u, v, sigma double-buffered)
  DevBuf<double> st[2];                  // [14*n]: u, v, 12 stresses
  struct View { double* p = nullptr; };
  View uv[2], sig[2];                    // views into st[k]: u at 0, v at n, stresses at 2n
  DevBuf<int32_t> iceumask;
  DevBuf<double> fm, strtltx, strtlty, strocnx, strocny, strintx, strinty;
  // out
  DevBuf<double> strairx, strairy, strength, divu, shear, rdg_conv, rdg_shear, prs_sig, strocnxT,
      strocnyT;
  // work (ice_dyn_evp.F90:170-184)
  DevBuf<double> tmass, umass, aiu, umassdtei, waterx, watery, forcex, forcey, work1;
  DevBuf<int32_t> icetmask;
  DevBuf<unsigned long long> counters;

  // captured subcycle loop (hipGraph), keyed by (cur, ksub0, nsub, tile shape)
  hipGraphExec_t graph_exec = nullptr;
  int graph_key[4] = {-1, -1, -1, -1};

  void launch_subcycle(int ksub);
  void launch_subcycle_pair(int ksub);
  void launch_subcycle_skew(int ksub, int K, bool flip_and_halo = true, hipStream_t on = nullptr);
  void skew_args(SkewArgs& sa, int K);
  void skew_launch(const SkewArgs& sa, int K, bool last, int nt, hipStream_t s);
  // the sweep in front of a wide-halo refresh as two launches: edge segments + refresh on the main stream, interior beside them
  bool split_on = false, in_capture = false;   // (off by default: on one GPU the two-launch form costs more than it hides)
  void build_split(int K);
  void launch_subcycle_skew_split(int ksub, int K);
  void launch_subcycle_skew_ext(int ksub, int K, int ext);
  bool trim_ext_on = true;
  int split_probe = 0;
  struct TileTab { int K, ext, S, nb, edge, total; bool split; DevBuf<int32_t> tab; };
  std::vector<std::unique_ptr<TileTab>> tile_tabs;
  const TileTab& tiles_for(int K, int ext, bool split);
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  void launch_range(int ksub0, int nsub);
  void after_subcycle(int ksub);
  SubArgs make_args() const;
  void drop_graph();
};

// host-only: is the sweep kernel's column layout right for a ring of ncol + 1 positions with this shift? (evp.hip: skew_layout_ok)
bool evp_skew_layout_ok(int K, int S, int ncol, int shift, bool cyc);

#ifdef CICE4_AMD_AUSCOM
// the coupled flavour's namelist variables (ice_dyn_evp.F90:91-97, ice_init.F90:258-264) on the current device
void evp_set_namelist(double cosw, double sinw, double dragio, int use_ocnslope);
#endif

}  // namespace cice
