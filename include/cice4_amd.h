/* cice4_amd -- C-ABI of the MI355X-native EVP-dynamics + column-thermodynamics
 * hot path of CICE4 (COSIMA/cice4).  Plain pointers and sizes only: this is
 * what a Fortran `bind(C)` interface block (cice4_amd/fortran/) binds to.
 *
 * Array convention (= the reference's): every field is a contiguous
 * column-major Fortran array, i fastest, dimensioned (nx_block,ny_block) or
 * (nx_block,ny_block,nblocks) with one ghost cell on each side
 * (source/ice_blocks.F90:56-62); 3-D/4-D fields put the extra dimension(s)
 * between (nx_block,ny_block) and the block index exactly as
 * source/ice_state.F90:55-148 does.  `logical` masks are 4-byte (non-zero =
 * .true.).  Index lists hold 1-based indices.  All reals are double.
 *
 * Every function returns 0 on success or a negative CICE_E* code; the text
 * is available from cice_last_error().  Nothing here ever calls exit()/abort():
 * physics failures come back as (l_stop, istop, jstop) for the caller's
 * abort_ice, as in drivers/cice4/CICE_RunMod.F90:532-544.
 *
 * Citations are file:line under the reference checkout.
 */
#ifndef CICE4_AMD_H
#define CICE4_AMD_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CICE_NCAT 5       /* source/ice_domain_size.F90:37-47 */
#define CICE_NILYR 4
#define CICE_NSLYR 1
#define CICE_MAX_NTRCR 5

enum {
  CICE_OK = 0,
  CICE_EINVAL = -1,   /* bad argument / call order */
  CICE_EDEVICE = -2,  /* HIP runtime error (no GPU, out of memory, launch failure) */
  CICE_ECOMM = -3,    /* RCCL error */
  CICE_EUNSUPPORTED = -4
};

typedef struct cice_ctx cice_ctx;

/* ---- lifecycle --------------------------------------------------------- */
/* One context = one rank = one GPU (the reference is one MPI task, no threads:
 * mpi/ice_communicate.F90:109-136).  device < 0: use HIP's current device. */
int cice_create(cice_ctx **ctx, int device);
/* number of HIP devices visible to this process (0 without a GPU): an MPI task picks my_task mod this */
int cice_device_count(void);
/* Page-lock a host array that will be handed to the drop-in entries again and again (the model's module
 * arrays keep their addresses for the whole run): transfers then run as asynchronous DMA instead of
 * through the runtime's staging buffer -- a latency matter, 71 separate 1-MB fields per evp(dt) at gx1.
 * Purely an optimisation: arrays that cannot be registered are used pageable.  Registered ranges are
 * released by cice_destroy. */
int cice_host_register(cice_ctx *ctx, void *host, size_t bytes);
/* Release every range page-locked through this context (cice_host_register, cice_evp_pin_fields); call it
 * BEFORE the host frees or re-allocates such an array. */
int cice_host_unregister_all(cice_ctx *ctx);
int cice_destroy(cice_ctx *ctx);
/* The category / layer / tracer strides of every module array are the compile-time sizes above
 * (source/ice_domain_size.F90:37-70).  A host model calls this from its init routines with ITS ncat, nilyr, nslyr,
 * max_ntrcr: CICE_EINVAL (and a message naming both sets) unless they are the library's. */
int cice_check_sizes(cice_ctx *ctx, int ncat, int nilyr, int nslyr, int max_ntrcr);
const char *cice_last_error(const cice_ctx *ctx); /* ctx may be NULL: last create error */
int cice_device_sync(cice_ctx *ctx);
/* Diagnostics: streams n_doubles 8-byte loads + stores through HBM twice (kernel
 * k_diag_copy8) and returns the second launch's time; used to calibrate the rocprofv3
 * FETCH_SIZE / WRITE_SIZE counters for the 8-byte-per-lane access width of the kernels. */
int cice_diag_stream_copy(cice_ctx *ctx, long long n_doubles, float *elapsed_ms);

/* ---- domain: replaces init_domain_blocks + init_domain_distribution +
 * ice_HaloCreate (source/ice_domain.F90:96,258; mpi/ice_boundary.F90:153).
 * Host logic only -- usable without a GPU. boundary: 0 open, 1 cyclic, 2 closed; ns_boundary also 3 =
 * 'tripole' (U-fold, source/ice_blocks.F90:457-467; needs a cyclic e-w boundary and an even nx_global) and 4 =
 * 'tripoleT' (fold through T points, serial/ice_boundary.F90:725-776: three rows of the top block row in the fold
 * buffer, which the update works on exactly as the serial reference leaves it -- its northeast / northwest messages
 * overwrite two of the three rows, :3813-3826 -- so that a run agrees with that reference bit for bit).
 * Blocks are dealt to an (npx x npy) process grid in contiguous rectangles
 * (cartesian distribution, source/ice_distribution.F90:78). */
int cice_domain_create(cice_ctx *ctx, int nx_global, int ny_global, int block_size_x,
                       int block_size_y, int ew_boundary, int ns_boundary, int rank, int npx,
                       int npy);
/* The same for ANY block->task map (what create_distribution produced: cartesian, rake, space curve --
 * source/ice_distribution.F90:78-190 -- with land-block elimination): owner[g] = task of global block g
 * (g = jblock * nblocks_x + iblock, 0-based) or -1 for an eliminated block, local_id[g] = its 0-based position
 * among that task's blocks (NULL: ascending g).  Ghost cells facing an eliminated block take the fill value. */
int cice_domain_create_map(cice_ctx *ctx, int nx_global, int ny_global, int block_size_x,
                           int block_size_y, int ew_boundary, int ns_boundary, int rank, int nranks,
                           const int *owner, const int *local_id);
/* Host copy of one index list of the domain: "hfill" (ghost cells facing eliminated blocks), and the tripole
 * lists "fold_lsrc"/"fold_bidx" (local top rows -> global buffer), "fold_lo"/"fold_hi" (symmetry pairs),
 * "fold_dst"/"fold_src" (copy out) for field location loc (1..4).  *n = length; out may be NULL. */
int cice_domain_list(const cice_ctx *ctx, const char *name, int loc, int *n, int32_t *out);
/* Wide-halo variant for j-slab decompositions (strong scaling over GPUs): nblocks_y slabs of
 * full width dealt to nranks ranks in contiguous runs; every slab is extended by `overlap`
 * rows into its neighbours.  The overlap rows are recomputed redundantly by the subcycle kernel
 * and refreshed from their owner (u, v and the 12 stresses in ONE message per neighbour) only
 * every `overlap` subcycles instead of after every subcycle: results on the owned rows are
 * bit-identical, the number of exchanges drops by that factor.  Host arrays then describe the
 * EXTENDED blocks (cice_domain_block gives the owned rows).  overlap = 0: plain slabs.
 * ns_boundary may be a tripole fold (3, 4; east-west cyclic, nx_global even, overlap <= slab height - 4): the rank with
 * the top slab folds u, v after every subcycle as on any tripole grid, everything else is as above. */
int cice_domain_create_slabs(cice_ctx *ctx, int nx_global, int ny_global, int nblocks_y,
                             int ew_boundary, int ns_boundary, int rank, int nranks, int overlap);
/* info: nx_block, ny_block, nblocks(local), nblocks_tot, n_local_copies, n_send_msgs,
 *       n_recv_msgs, n_send_elems, n_recv_elems */
int cice_domain_info(const cice_ctx *ctx, int info[9]);
/* info: ilo, ihi, jlo, jhi (1-based, = type block, source/ice_blocks.F90:32-45),
 *       i0, j0 (0-based global index of cell ilo/jlo), global block id, owner rank,
 *       own_jlo, own_jhi (rows owned by the block; = jlo, jhi without overlap) */
int cice_domain_block(const cice_ctx *ctx, int local_block, int info[10]);
/* on-rank part of the wide-halo refresh (empty without overlap): n pairs */
int cice_domain_halo_refresh(const cice_ctx *ctx, int *n, int32_t *src, int32_t *dst);
/* on-rank halo copy list (0-based linear addresses into the (nx_block,ny_block,nblocks)
 * array): a[dst[n]] = a[src[n]]; = srcLocalAddr/dstLocalAddr of serial/ice_boundary.F90:49-62 */
int cice_domain_halo_local(const cice_ctx *ctx, int32_t *src, int32_t *dst);
/* msg-th send (dir=0) / recv (dir=1) message: peer rank, element count, addresses; dir = 2 / 3: the tripole
 * top-row messages (send: local addresses, receive: indices into the global fold buffer).  CICE_EINVAL past the last. */
int cice_domain_halo_msg(const cice_ctx *ctx, int dir, int msg, int *peer, int *count,
                         int32_t *addr);

/* ---- inter-GPU halo transport (RCCL over xGMI; replaces mpi/ice_boundary.F90's
 * MPI_ISEND/IRECV).  uid: 128-byte ncclUniqueId from cice_comm_unique_id on rank 0,
 * distributed by the caller (MPI_Bcast in a Fortran driver, torch.distributed here). */
int cice_comm_unique_id(char uid[128]);
int cice_comm_init(cice_ctx *ctx, const char uid[128], int rank, int nranks);
/* In-process stand-in for the communicator: the ranks of link `link_id` are contexts of THIS process (one host thread
 * each, on one GPU or several); messages travel device -> host mailbox -> device.  The counterpart of the reference's
 * serial/ directory next to mpi/: the whole multi-rank path runs on a box with one GPU (RCCL refuses two ranks on one
 * device).  Tests only -- every message synchronises the stream. */
int cice_comm_init_local(cice_ctx *ctx, int link_id, int rank, int nranks);
/* The same between PROCESSES of one host: a file `name` ("/...") under /dev/shm holds one mailbox of box_bytes per pair of
 * ranks; rank 0 creates it, the others wait for it.  For running a multi-process job (bench.py --gpus N with all ranks on
 * one device, the MPI build of the Fortran driver) on a box with one GPU.  Tests only. */
int cice_comm_init_shm(cice_ctx *ctx, const char *name, int rank, int nranks, long long box_bytes);
/* TIMING AID, not a communicator: this context is rank `rank` of an `nranks`-rank decomposition and is ALONE -- every
 * message it would send to a neighbour comes back as the message it would receive from that neighbour, device to device.
 * The rank runs the kernels, tile lists, pack / unpack and launch sequence of the nranks-rank job with nobody else on the
 * chip: what one GPU of a node would spend per subcycle, minus the link (no counterpart in the reference; DESIGN.md
 * section 7, bench.py --as-rank).  The fields it computes are those of a mirror boundary and mean nothing. */
int cice_comm_init_mirror(cice_ctx *ctx, int rank, int nranks);
/* ranks of the communicator as RCCL counts them (ncclCommCount; = MPI_COMM_SIZE of mpi/ice_communicate.F90:109-136);
 * 0 before cice_comm_init */
int cice_comm_count(cice_ctx *ctx, int *nranks);

/* ---- EVP dynamics (source/ice_dyn_evp.F90) ----------------------------- */
typedef struct { /* source/ice_grid.F90:58-133; each (nx_block,ny_block,nblocks) */
  const double *dxt, *dyt, *dxhy, *dyhx, *cxp, *cyp, *cxm, *cym, *tarea, *uarea, *tarear,
      *uarear, *tinyarea;
  const double *fcor;           /* fcor_blk, ice_dyn_evp.F90:105,503 */
  const int32_t *tmask, *umask; /* logical */
  /* Optional (may be NULL): the primary cell lengths HTN, HTE (ice_grid.F90:60-61).  When given and
   * the nine T-cell metrics above are bit-for-bit the functions of HTN/HTE that init_grid2 computes
   * (checked on the host at init), the subcycle kernel recomputes them instead of loading them. */
  const double *HTN, *HTE;
} cice_evp_grid;

typedef struct { /* namelist + ridging switches read by evp (ice_dyn_evp.F90:64-74; ice_mechred.F90:64-79) */
  int ndte, evp_damping;
  int kstrength, krdg_partic, krdg_redist;
  double mu_rdg;
} cice_evp_config;

/* init_evp (ice_dyn_evp.F90:441): allocates device state, uploads the grid, zeroes
 * uvel/vvel/stresses/iceumask on the device.  Requires cice_domain_create. */
int cice_evp_init(cice_ctx *ctx, const cice_evp_config *cfg, const cice_evp_grid *grid);

/* The module arrays evp(dt) reads and writes (ice_state.F90:55-148, ice_flux.F90:42-99).
 * All host pointers, (nx_block,ny_block,nblocks) unless noted.  in = read by evp,
 * io = read and written, out = written. */
typedef struct {
  /* in */
  const double *aice, *vice, *vsno, *aice0;
  const double *aicen, *vicen; /* (nx_block,ny_block,ncat,nblocks) */
  const double *strairxT, *strairyT, *uocn, *vocn, *ss_tltx, *ss_tlty;
  /* io */
  double *uvel, *vvel;
  double *stressp_1, *stressp_2, *stressp_3, *stressp_4, *stressm_1, *stressm_2, *stressm_3,
      *stressm_4, *stress12_1, *stress12_2, *stress12_3, *stress12_4;
  int32_t *iceumask; /* logical */
  double *fm, *strtltx, *strtlty, *strocnx, *strocny, *strintx, *strinty;
  /* out */
  double *strairx, *strairy, *strength, *divu, *shear, *rdg_conv, *rdg_shear, *prs_sig,
      *strocnxT, *strocnyT;
} cice_evp_fields;
/* cice_host_register for every array of the struct (once, for arrays that keep their addresses) */
int cice_evp_pin_fields(cice_ctx *ctx, const cice_evp_fields *f);

/* Drop-in for `call evp(dt)` (ice_dyn_evp.F90:119-432): upload, run on the GPU,
 * download.  Same results as cice_evp_upload + cice_evp_step + cice_evp_download, as one pipeline: the six fields the
 * preparation leaves final (strairx, strairy, strength, fm, strtltx, strtlty) and iceumask travel to the host while the
 * subcycle loop runs (on a domain of several ranks they follow the loop with the rest: ranks that share one device in the
 * rehearsals need their hardware queues for the loops that wait for each other).
 * Two statements a caller may make about itself (cice_evp_set_option; both off by default, both hold for the reference's
 * unchanged driver):
 *   "keep_state" = 1: between two cice_evp calls the caller does not change uvel, vvel, the 12 stresses or iceumask on
 *     the host (in the reference only evp itself writes them, and the restart reader before the first step): from the
 *     second call on they are not uploaded -- the device copies the last call left are the input.  = 2: and fm,
 *     strtltx/y, strocnx/y, strintx/y (which evp reads back outside its ice mask) are ZERO on the host when cice_evp is
 *     called, as init_history_dyn leaves them at the top of every step (ice_flux.F90:585-602): zeroed on the device.
 *     cice_evp_init, cice_evp_upload / _prepare / _subcycles / _finish / _step end the statement's effect for one call.
 *   "lazy_stresses" = 1: cice_evp leaves the 12 stresses on the device; cice_evp_download_stresses brings them to the host
 *     (the reference reads them there only for its history and restart files: ice_history.F90:1939, ice_restart.F90:74-256).
 * gx1 size over PCIe (profiles/r05_bench_gx1.json, pcie_inclusive): see DESIGN.md section 6. */
int cice_evp(cice_ctx *ctx, double dt, cice_evp_fields *f);
int cice_evp_download_stresses(cice_ctx *ctx, cice_evp_fields *f);   /* the 12 stresses of the device state -> host */
/* Device-resident hand-off from the thermodynamic half-step to the dynamics (SURVEY section 8 f1;
 * drivers/cice4/CICE_RunMod.F90:374-591 -> source/ice_step_mod.F90:575): aicen, vicen of the batched thermo state
 * (cice_step_therm1 / cice_thermo_batch_step leave them on the device) become the dynamics' aicen, vicen, and aice, vice,
 * vsno, aice0 are aggregated from them on the device in the order of `aggregate` (source/ice_itd.F90:279).  The NEXT
 * cice_evp / cice_evp_upload may then pass NULL for those six fields.  For a caller that has not changed the state on
 * the host in between (the reference's step_therm2 / transport do: its unchanged driver keeps uploading). */
int cice_evp_adopt_thermo_state(cice_ctx *ctx);
int cice_evp_upload(cice_ctx *ctx, const cice_evp_fields *f);   /* in + io fields -> HBM */
int cice_evp_step(cice_ctx *ctx, double dt);                    /* evp(dt) on resident state */
int cice_evp_download(cice_ctx *ctx, cice_evp_fields *f);       /* io + out fields -> host */

/* Pieces of evp on the resident state, for tests and measurement:
 * prepare = ice_dyn_evp.F90:214-344 (prep1, masks, T->U, prep2, strength, halos);
 * subcycles = nsub passes of :347-404 starting at subcycle ksub0 (1-based; the pass with
 * ksub == ndte also writes divu, shear, rdg_conv, rdg_shear, prs_sig, strint, strocn); finish = :410-428.
 * elapsed_ms (may be NULL) is the HIP-event time of the launches on the library's stream; where the range runs as the one-launch
 * loop, of that ONE launch (what rocprofv3 reports for the kernel), without the small launch that picks the tile map in front
 * of it and the read-back of the abort word behind it. */
int cice_evp_prepare(cice_ctx *ctx, double dt);
int cice_evp_subcycles(cice_ctx *ctx, int ksub0, int nsub, float *elapsed_ms);
int cice_evp_finish(cice_ctx *ctx);
/* tuning / A-B switches: "waves" (wavefronts per workgroup: 4, 8, 16), "rows_per_wave" (T-rows
 * per wavefront: 1, 2, 4, 8), "use_graph" (0/1), "derive_metrics" (0/1, see cice_evp_grid),
 * "fuse" (0/1: two subcycles per launch where no ghost row of a local block changes between
 * subcycles -- one full-width block per rank or wide-halo slabs with an even overlap),
 * "fused_waves" (0 = auto, 8, 12, 13, 14, 16), "resident" (0/1, default 1: the whole subcycle loop in ONE launch with
 * the state in registers -- one block per rank, no inter-rank or tripole exchange during the subcycling, at most one
 * 64 x W tile per compute unit (gx3, gx1; not 0.1 degree); 2 = on and forget an earlier time-out; environment
 * CICE4_AMD_RESIDENT=0 switches it off), "resident_waves" (0 = auto, 4, 6, 8, 11, 12), "resident_dense" (0/1, default 1:
 * three 4-wavefront workgroups per compute unit where that fills the chip exactly -- gx1 -- so that hand-offs of one
 * overlap the arithmetic of the others; a launch that does not get every slot times out and later ones use one workgroup
 * per compute unit), "resident_spin_us" (default 200000: how long a tile of the resident loop waits for a neighbour
 * before the launch gives up and the range is run by the launch-per-pair loop; 0 makes every wait fail -- tests),
 * "resident_map" (-1 default: which tile a workgroup of that loop takes is chosen on the device, once per evp(dt), by the ice
 * cover -- under ice in latitude bands a CU gets one tile with ice and two without; 0 / 1 fix the map).
 * Sweeps (K subcycles per launch on grids of ~0.1 degree size): "skew" (0/1), "skew_levels" (K: 2, 3, 4 default; 5, 6, 8 in -DCICE4_AMD_EXPERIMENTS builds),
 * "skew_min_cells", "skew_seg_rows" (rows per workgroup, 0 = as many workgroups as the chip holds), "skew_rowact" (0/1,
 * default 1: a workgroup walks only the runs of rows of its segment that hold ice), "skew_balance" (0/1, default 1 on
 * one-block domains: the segment table is re-cut from the workgroups' measured times -- the sweeps of the first loop after
 * start-up are measured, eagerly; "skew_balance_every", default 96: loops between two later tuning phases of 8 sweeps.
 * WHAT THIS COSTS: a measured sweep is launched without the captured graph and followed by one hipStreamSynchronize plus one
 * blocking read-back of two clock words per workgroup -- the first loop after a new table (36 sweeps) takes about 1.7 x its
 * later time, and one loop in every 96 runs its first 8 sweeps that way; the segment table then depends on measured times,
 * so the launch geometry -- never a result -- differs from run to run.  0 keeps the static table and replays the graph always),
 * "resident_granules" (one-launch loop on one rank: 0 progress words, 1 data-tagged granules in a free-running loop unless the
 * last step's ice cover left most tiles empty, 2 always),
 * "skew_fill" / "skew_gen_pct" (per cent: static weights of a workgroup's place -- more rows on a CU that holds fewer
 * workgroups, more for the workgroup dispatched first; defaults 26 / 10), "skew_split" (wide-halo slabs: the refresh beside
 * the interior sweep), "skew_subs" (1; 3 wavefronts per level in -DCICE4_AMD_EXPERIMENTS builds).  DESIGN.md sections 3.1, 3.2, 7.
 * Results never depend on them; cice_evp_init picks waves / rows_per_wave from the grid size.
 * cice_evp_get_info keys: "derive_metrics" (1 if active), "waves", "rows_per_wave", "fused"
 * (1 if this domain runs two subcycles per launch), "fused_waves", "resident" (1 if the next cice_evp_subcycles
 * of two or more subcycles runs as one launch), "resident_waves", "resident_dense" (1 if with several workgroups per
 * compute unit), "resident_granules" (1 if with the granule hand-off), "resident_map" (the map last chosen, -1 before the first loop), "skew" / "skew_fold" (1 if sweeps apply),
 * "skew_levels", "skew_strips", "skew_seg_rows", "skew_rowact", "skew_balance", "skew_balanced" (sweeps measured so far),
 * "skew_fill", "skew_pairs", "skew_subs", "skew_split", "skew_trim_ext", "last_launches" (kernel launches of the last
 * subcycle range: 1 = the one-launch loop). */
int cice_evp_set_option(cice_ctx *ctx, const char *key, int value);
int cice_evp_get_info(cice_ctx *ctx, const char *key, int *value);
/* number of T-cells with icetmask = 1 and U-cells with iceumask on this rank after prepare
 * (= sum of icellt / icellu, ice_dyn_evp.F90:160-162) */
int cice_evp_active_cells(cice_ctx *ctx, long long *n_tcells, long long *n_ucells);
/* The one-launch subcycle loop ACROSS RANKS: ONE BLOCK PER RANK in any cartesian layout -- full-width j-slabs, the i-slabs
 * of bld/config.nci.access-om.360x300 (6 x 1 tasks), the 2 x 2 tasks of comp_ice:34-46; source/ice_blocks.F90:133-330 --
 * with up to eight neighbouring ranks (round 5; before: j-slabs only).  The tiles on
 * a block's edges exchange their edge velocities with the neighbouring rank's tiles through stores into the
 * neighbour's exchange copies and progress words (what replaces the two ice_HaloUpdate calls per subcycle,
 * source/ice_dyn_evp.F90:397-402, mpi/ice_boundary.F90:1028-1417: no message, no host involvement inside the loop).
 * Every rank exports its buffers (cice_evp_peer_export: device pointers xu0, xu1, progress words, and its plane size),
 * hands them to its neighbours through its control plane and connects what it receives: cice_evp_peer_ranks lists the
 * ranks this rank's block exchanges ghost cells with (ascending; diagonal neighbours and the ones across a cyclic edge
 * included), cice_evp_peer_connect_rank connects one of them.  (The older cice_evp_peer_connect takes a side instead, for
 * j-slabs: 0 = the rank to the south, 1 = to the north.)  Two contexts of one process on one GPU pass the pointers as they are (tests); processes on
 * different GPUs use the _ipc forms (hipIpcGetMemHandle / hipIpcOpenMemHandle; handles travel like the ncclUniqueId).
 * With every neighbour connected, cice_evp_get_info("resident_peer") is 1 and cice_evp / cice_evp_subcycles run the loop
 * as one launch per rank; a rank whose launch times out raises a flag that is all-reduced over the communicator
 * ("resident_peer_agree", default 1) so that all ranks fall back to the launch-per-pair loop together.
 * Under a tripole north boundary (ns_boundary 3, 4) the loop runs where every rank holds ONE block that spans the width
 * (j-slabs): the rank with the top slab carries the fold inside its loop, as a one-rank domain does
 * (serial/ice_boundary.F90:705-869 on the degenerate row); layouts with a rank boundary through the fold keep the
 * message path (resident_peer stays 0).
 * "resident_peer_share": contexts sharing one device (default 1; the one-GPU test uses 2).
 * Memory types: the three exported buffers -- what another device writes (edge velocities, progress words) while a launch
 * of this one polls and reads them -- are FINE-GRAINED device memory (hipExtMallocWithFlags(hipDeviceMallocFinegrained));
 * every access to them that crosses devices is system-scope.  cice_evp_get_info("resident_peer_fine") says so. */
int cice_evp_peer_export(cice_ctx *ctx, void *bufs[3], long long *plane);
int cice_evp_peer_connect(cice_ctx *ctx, int side, void *xu0, void *xu1, void *rprog, long long plane);
int cice_evp_peer_export_ipc(cice_ctx *ctx, char handles[3][64], long long *plane);
int cice_evp_peer_connect_ipc(cice_ctx *ctx, int side, const char handles[3][64], long long plane);
int cice_evp_peer_ranks(cice_ctx *ctx, int *n, int32_t ranks[8]);
int cice_evp_peer_connect_rank(cice_ctx *ctx, int rank, void *xu0, void *xu1, void *rprog, long long plane);
int cice_evp_peer_connect_rank_ipc(cice_ctx *ctx, int rank, const char handles[3][64], long long plane);
/* Test / tuning aid: `what` = "skew_times" (after cice_evp_set_option("skew_debug", 1)): start and end wall-clock ticks
 * (10 ns) of every workgroup of the last K-subcycle sweep launch; "stamps" (after cice_evp_set_option("stamps", 1), in a
 * DIAGNOSTIC build of the library compiled -DCICE4_AMD_STAMPS only -- the product build's kernels hold no stamp and the
 * array comes back zero): per workgroup of the last one-launch loop / sweep {cycles before, cycles after, 100 MHz ticks
 * before, ticks after} its loop (s_memtime / s_memrealtime: the in-kernel clock, scripts/inkernel_clock.py); "skew_rows":
 * the sweep's segment table as it stands ([segments][strips][2]: first / last U-row of a workgroup relative to the block's
 * first row) -- the library re-cuts it from measured workgroup times (option "skew_balance", on by default on one-block
 * domains; cice_evp_get_info "skew_balance" / "skew_balanced" = sweeps measured so far).
 * *count: in = capacity of out (out may be NULL), out = entries available. */
int cice_evp_debug(cice_ctx *ctx, const char *what, long long *out, long long *count);

/* Per-routine entries with the reference's own argument lists (host pointers, one
 * (nx_block,ny_block) block), for parity tests: stress ice_dyn_evp.F90:947-966,
 * stepu :1302-1313.  Scalars come from the cice_evp_config/dt given here. */
int cice_evp_stress(cice_ctx *ctx, double dt, int ndte, int evp_damping, int nx_block,
                    int ny_block, int ksub, int icellt, const int32_t *indxti,
                    const int32_t *indxtj, const double *uvel, const double *vvel,
                    const double *dxt, const double *dyt, const double *dxhy, const double *dyhx,
                    const double *cxp, const double *cyp, const double *cxm, const double *cym,
                    const double *tarear, const double *tinyarea, const double *strength,
                    double *stressp_1, double *stressp_2, double *stressp_3, double *stressp_4,
                    double *stressm_1, double *stressm_2, double *stressm_3, double *stressm_4,
                    double *stress12_1, double *stress12_2, double *stress12_3,
                    double *stress12_4, double *shear, double *divu, double *prs_sig,
                    double *rdg_conv, double *rdg_shear, double *str /* (nx,ny,8) */);
int cice_evp_stepu(cice_ctx *ctx, int nx_block, int ny_block, int icellu, const int32_t *indxui,
                   const int32_t *indxuj, const double *aiu, const double *str,
                   const double *uocn, const double *vocn, const double *waterx,
                   const double *watery, const double *forcex, const double *forcey,
                   const double *umassdtei, const double *fm, const double *uarear,
                   double *strocnx, double *strocny, double *strintx, double *strinty,
                   double *uvel, double *vvel);

/* Generic halo update of a host field through the device path (ice_HaloUpdate2DR8 /
 * 2DI4, mpi/ice_boundary.F90:1028,1820): nlev slabs of (nx_block,ny_block,nblocks). */
int cice_halo_update_r8(cice_ctx *ctx, double *field, int nlev);
int cice_halo_update_i4(cice_ctx *ctx, int32_t *field, int nlev);
/* The same for a field that is RESIDENT in device memory (ice_HaloUpdate3DR8 / 4DR8 with all levels in one
 * message per neighbour, mpi/ice_boundary.F90:2216,3587; bound_state's 65 levels, source/ice_state.F90:162-217):
 * nlev planes of nx_block*ny_block*nblocks elements; asynchronous on the library's stream (cice_device_sync). */
int cice_halo_update_dev_r8(cice_ctx *ctx, double *dev_field, int nlev);
int cice_halo_update_dev_i4(cice_ctx *ctx, int32_t *dev_field, int nlev);
/* With the field location and kind of ice_HaloUpdate (ice_constants.F90:185-205: 1 center, 2 NE corner, 3 N face,
 * 4 E face; 1 scalar, 2 vector, 3 angle) -- they set offsets and sign at a tripole fold
 * (serial/ice_boundary.F90:705-869) -- and its fillValue for ghost cells that face an eliminated land block
 * (mpi/ice_boundary.F90:5108-5111).  R4 fields are handled in single precision as the reference does. */
int cice_halo_update_ex_r8(cice_ctx *ctx, double *field, int nlev, int loc, int kind, double fill);
int cice_halo_update_ex_r4(cice_ctx *ctx, float *field, int nlev, int loc, int kind, float fill);
int cice_halo_update_ex_i4(cice_ctx *ctx, int32_t *field, int nlev, int loc, int kind, int32_t fill);
/* ... and for a host field in the array layout ice_HaloUpdate3D/4D receive, (nx_block, ny_block, nz, nblocks) with the
 * block index last (nz = product of the level dimensions; the task's blocks are the first ones): no repacking on the
 * host, strided copies to the device's level-major layout instead. */
int cice_halo_update_blocked_r8(cice_ctx *ctx, double *field, int nz, int loc, int kind, double fill);
int cice_halo_update_blocked_r4(cice_ctx *ctx, float *field, int nz, int loc, int kind, float fill);
int cice_halo_update_blocked_i4(cice_ctx *ctx, int32_t *field, int nz, int loc, int kind, int32_t fill);
/* a section of a 4-d module array such as trcrn(:,:,1:ntrcr,:,:) (ice_state.F90:206): whole horizontal planes, level
 * (z1, z2) of local block b at field + b * stride_block + z2 * stride2 + z1 * stride1 (strides in elements) */
int cice_halo_update_strided_r8(cice_ctx *ctx, double *field, int nz1, long long stride1, int nz2, long long stride2,
                                long long stride_block, int loc, int kind, double fill);
int cice_halo_update_dev_ex_r8(cice_ctx *ctx, double *dev_field, int nlev, int loc, int kind, double fill);
/* Device memory on the context's GPU for such resident fields, and blocking copies ordered on the library's stream. */
int cice_device_alloc(cice_ctx *ctx, size_t bytes, void **dev);
int cice_device_free(cice_ctx *ctx, void *dev);
int cice_device_copy(cice_ctx *ctx, void *dst, const void *src, size_t bytes, int to_device);

/* ---- column thermodynamics (source/ice_therm_vertical.F90) -------------- */
typedef struct { /* module variables :56-79 + tracer slots (ice_state.F90 nt_Tsfc, nt_iage) */
  int heat_capacity, calc_Tsfc; /* logical */
  int conduct;                  /* 0 = 'MU71', 1 = 'bubbly' */
  double ustar_min;
  int tr_iage, nt_Tsfc, nt_iage; /* 1-based */
} cice_thermo_config;

/* init_thermo_vertical (:533): salinity / melting-temperature profile; returns them
 * (nilyr+1 values each) when the pointers are non-NULL. */
int cice_thermo_init(cice_ctx *ctx, const cice_thermo_config *cfg, double *salin, double *Tmlt);

/* Drop-in for `call thermo_vertical(...)` (:108-132), same argument order, host pointers;
 * trcrn is (nx,ny,max_ntrcr), eicen (nx,ny,nilyr), esnon (nx,ny,nslyr). */
int cice_thermo_vertical(cice_ctx *ctx, int nx_block, int ny_block, double dt, int icells,
                         const int32_t *indxi, const int32_t *indxj, double *aicen,
                         double *trcrn, double *vicen, double *vsnon, double *eicen,
                         double *esnon, const double *flw, const double *potT, const double *Qa,
                         const double *rhoa, const double *fsnow, const double *fbot,
                         const double *Tbot, const double *lhcoef, const double *shcoef,
                         double *fswsfc, double *fswint, double *fswthrun, double *Sswabs,
                         double *Iswabs, double *fsurfn, double *fcondtopn, double *fsensn,
                         double *flatn, double *fswabsn, double *flwoutn, double *evapn,
                         double *freshn, double *fsaltn, double *fhocnn, double *meltt,
                         double *melts, double *meltb, double *congel, double *snoice,
                         double *mlt_onset, double *frz_onset, double yday, int32_t *l_stop,
                         int32_t *istop, int32_t *jstop);

/* Batched, device-resident form: every category of every local block in one launch
 * (the n = 1..ncat loop of step_therm1, drivers/cice4/CICE_RunMod.F90:374-591, around
 * thermo_vertical; cells with aicen(i,j,n) > puny on the physical domain are updated).
 * Field shapes: state aicen/vicen/vsnon (nx,ny,ncat,nb), trcrn (nx,ny,max_ntrcr,ncat,nb),
 * eicen (nx,ny,ncat*nilyr,nb), esnon (nx,ny,ncat*nslyr,nb); forcing (nx,ny,nb);
 * per-category fields (nx,ny,ncat,nb); Sswabs (nx,ny,nslyr,ncat,nb); Iswabs (nx,ny,nilyr,ncat,nb). */
typedef struct {
  double *aicen, *trcrn, *vicen, *vsnon, *eicen, *esnon;
  const double *flw, *potT, *Qa, *rhoa, *fsnow, *fbot, *Tbot;
  const double *lhcoef, *shcoef;                      /* per category */
  double *fswsfc, *fswint, *fswthrun, *Sswabs, *Iswabs; /* per category */
  double *fsurfn, *fcondtopn, *fsensn, *flatn, *fswabsn, *flwoutn, *evapn, *freshn, *fsaltn,
      *fhocnn, *meltt, *melts, *meltb, *congel, *snoice; /* per category, out */
  double *mlt_onset, *frz_onset;                         /* (nx,ny,nb) io */
} cice_thermo_fields;

int cice_thermo_batch_alloc(cice_ctx *ctx, int nx_block, int ny_block, int nblocks);
/* Host -> device / device -> host of the batch.  Of trcrn(nx, ny, max_ntrcr, ncat, nblocks) only the surface-temperature
 * plane (nt_Tsfc of cice_thermo_init) of every (category, block) travels: the column physics reads and writes no other
 * tracer; the other planes of the host array are left as they are. */
int cice_thermo_batch_upload(cice_ctx *ctx, const cice_thermo_fields *host);
/* Tuning switches of the batched step; results never depend on them.  "sort_chunk" (0 = off, the default; 256 .. 2048):
 * the columns of every chunk of that many consecutive cells of a (category, block) plane are ordered by the work they
 * are expected to take (the solver iterations of the previous step, snow / no snow, cold / melting surface) before
 * wavefronts are formed, "sort_group" (1, 2, 4, 8, 16, 32) adjacent cells staying together (built,
 * bit-exact, measured slower than the unsorted kernel at every setting, hence off). */
int cice_thermo_set_option(cice_ctx *ctx, const char *key, int value);
/* One pass over all (cell,category) columns.  n_updates: number of columns updated;
 * l_stop/istop/jstop/nstop/bstop: first failing column in (block, category, list) order. */
int cice_thermo_batch_step(cice_ctx *ctx, double dt, double yday, long long *n_updates,
                           int32_t *l_stop, int32_t *istop, int32_t *jstop, int32_t *nstop,
                           int32_t *bstop, float *elapsed_ms);
int cice_thermo_batch_download(cice_ctx *ctx, cice_thermo_fields *host);

/* merge_fluxes (source/ice_flux.F90:613-762) for all categories of all blocks after a batched
 * step: acc[k] += catn[k] * aicen_init over n = 1..ncat in category order, on the cells of each
 * category's list (aicen_init(i,j,n) > puny on the physical domain), flwout with its
 * -(1-emissivity)*flw term.  The 15 thermo outputs and fswthrun come from the device-resident
 * batch; strairxn, strairyn, Trefn, Qrefn (atmo_boundary_layer outputs, (nx,ny,ncat,nb)) and
 * aicen_init are given here.  acc[20]: host (nx,ny,nb) cumulative fields in the order
 * strairxT, strairyT, fsurf, fcondtop, fsens, flat, fswabs, flwout, evap, Tref, Qref, fresh, fsalt,
 * fhocn, fswthru, meltt, meltb, melts, congel, snoice; updated in place. */
typedef struct {
  const double *aicen_init, *strairxn, *strairyn, *Trefn, *Qrefn;
  double *acc[20];
} cice_merge_fields;
int cice_thermo_batch_merge(cice_ctx *ctx, const cice_merge_fields *f);

/* The thermodynamic half of a time step in ONE call (SURVEY section 8(b) `cice_step_therm1_block`; what
 * step_therm1, drivers/cice4/CICE_RunMod.F90:260-598, does per block and category): one upload of the state
 * (cice_thermo_fields, fbot/Tbot ignored), frzmlt_bottom_lateral (:363) on the device from `fz`,
 * thermo_vertical for every category of every block (:502), merge_fluxes (:565; mg->aicen_init may be NULL: the
 * concentrations before the update are kept on the device), one download (state, per-category outputs, merged
 * accumulators, Tbot / fbot / rside where non-NULL).  atmo_boundary_layer (:402) stays with the caller: its
 * per-category outputs lhcoef, shcoef (st) and strairxn, strairyn, Trefn, Qrefn (mg) are inputs.
 * All fields (nx,ny[,..],nb) as in cice_thermo_fields; needs cice_thermo_batch_alloc. */
typedef struct {
  const double *aice, *frzmlt, *sst, *Tf, *strocnxT, *strocnyT; /* (nx,ny,nb) in */
  double *Tbot, *fbot, *rside;                                  /* (nx,ny,nb) out, may be NULL */
} cice_frzmlt_fields;
int cice_step_therm1(cice_ctx *ctx, double dt, double yday, cice_thermo_fields *state,
                     const cice_frzmlt_fields *fz, const cice_merge_fields *mg, long long *n_updates,
                     int32_t *l_stop, int32_t *istop, int32_t *jstop, int32_t *nstop, int32_t *bstop);

/* ... and atmo_boundary_layer (source/ice_atmo.F90:56-384; CICE_RunMod.F90:402-425) on the device too, in front of
 * thermo_vertical: for every category the cells with aicen > puny, Tsf = trcrn(:,:,nt_Tsfc,n,iblk) before the column
 * update.  state->lhcoef / shcoef and mg->strairxn / strairyn / Trefn / Qrefn are then NOT read (may be NULL); what
 * the routine produced is handed back through the six output pointers below where they are non-NULL
 * ((nx,ny,ncat,nb), zero outside a category's cells).  calc_strair = 0: strairxn/yn = strax/stray (:435-439).
 * exp is glibc's; log and atan are the device library's (<= 1 ulp), so this stage agrees with the reference to
 * ~1e-13, not bit for bit (tests/test_gpu_atmo.py) -- the bit-exact configuration is cice_step_therm1. */
typedef struct {
  const double *uatm, *vatm, *wind, *zlvl; /* (nx,ny,nb) in */
  const double *strax, *stray;             /* (nx,ny,nb) in, calc_strair = 0 only */
  int calc_strair;
  double *strairxn, *strairyn, *Trefn, *Qrefn, *lhcoef, *shcoef; /* (nx,ny,ncat,nb) out, may be NULL */
} cice_atmo_fields;
int cice_step_therm1_abl(cice_ctx *ctx, double dt, double yday, cice_thermo_fields *state,
                         const cice_frzmlt_fields *fz, const cice_merge_fields *mg, const cice_atmo_fields *atm,
                         long long *n_updates, int32_t *l_stop, int32_t *istop, int32_t *jstop, int32_t *nstop,
                         int32_t *bstop);

/* atmo_boundary_layer (source/ice_atmo.F90:56-384) with the reference's argument list, one block, host pointers:
 * sfctype 0 = 'ice', 1 = 'ocn'; calc_strair = the module variable of ice_atmo (0: strx, stry are left untouched).
 * All arrays (nx_block,ny_block); every output is zero outside the list. */
int cice_atmo_boundary_layer(cice_ctx *ctx, int nx_block, int ny_block, int sfctype, int icells,
                             const int32_t *indxi, const int32_t *indxj, const double *Tsf, const double *potT,
                             const double *uatm, const double *vatm, const double *wind, const double *zlvl,
                             const double *Qa, const double *rhoa, int calc_strair, double *strx, double *stry,
                             double *Tref, double *Qref, double *delt, double *delq, double *lhcoef,
                             double *shcoef);

/* frzmlt_bottom_lateral (:605-824), one block, host pointers;
 * eicen (nx,ny,ntilyr), esnon (nx,ny,ntslyr). */
int cice_frzmlt_bottom_lateral(cice_ctx *ctx, int nx_block, int ny_block, int ilo, int ihi,
                               int jlo, int jhi, double dt, const double *aice,
                               const double *frzmlt, const double *eicen, const double *esnon,
                               const double *sst, const double *Tf, const double *strocnxT,
                               const double *strocnyT, double *Tbot, double *fbot,
                               double *rside);

/* ---- horizontal transport by incremental remapping ----------------------------------
 * cice_transport_init ≙ init_transport (source/ice_transport_driver.F90:81; advection = 'remap'): tracer
 * dependencies from ntrcr / trcr_depend (ice_init.F90:848-852: 0 area tracer, 1 ice-volume, 2 snow-volume tracer),
 * one upload of the grid arrays of ice_grid (HTN, HTE, dxt, dyt, dxu, dyu, tarear, hm; (nx_block,ny_block,nblocks)).
 * cice_transport_remap ≙ `call transport_remap(dt)` (ice_step_mod.F90:600; driver :179-663 with
 * l_conservation_check = l_monotonicity_check = F): state_to_tracers, horizontal_remap (make_masks,
 * construct_fields, departure_points, locate_triangles, triangle_coordinates, transport_integrals, update_fields),
 * tracers_to_state and bound_state, all on the device; host arrays in the reference's layout: aice0, uvel, vvel
 * (nx,ny,nb), aicen/vicen/vsnon (nx,ny,ncat,nb), trcrn (nx,ny,max_ntrcr,ncat,nb), eicen (nx,ny,ntilyr,nb),
 * esnon (nx,ny,ntslyr,nb).  l_stop: 0 ok, 1 departure point outside the neighbouring cells, 2 negative area
 * (the caller's abort_ice), with a cell (istop, jstop).  Compile-time choices of the reference kept:
 * l_fixed_area = F, integral_order = 3, l_dp_midpt = T.  advection = 'upwind' is not provided. */
typedef struct {
  int ntrcr;
  int trcr_depend[CICE_MAX_NTRCR];
} cice_transport_config;
typedef struct {
  const double *HTN, *HTE, *dxt, *dyt, *dxu, *dyu, *tarear, *hm;
} cice_transport_grid;
typedef struct {
  double *aice0, *aicen, *trcrn, *vicen, *vsnon, *eicen, *esnon;
  const double *uvel, *vvel;
} cice_transport_fields;
int cice_transport_init(cice_ctx *ctx, const cice_transport_config *cfg, const cice_transport_grid *grid);
int cice_transport_remap(cice_ctx *ctx, double dt, const cice_transport_fields *f, int32_t *l_stop,
                         int32_t *istop, int32_t *jstop);
/* Test aid, needs no device: the column layout of the K-subcycle sweep kernel for a block of ncol columns (K time levels,
 * S = 1 or 3 wavefronts per level; cyclic: east-west boundary).  Returns the number of lanes the strip at the ring's seam
 * gives up so that every owned column comes out right (checked by a lane-level restatement of the kernel's dependencies),
 * -1 if there is none; *strips = column strips of the block. */
int cice_debug_skew_layout(int K, int S, int ncol, int cyclic, int *strips);
/* Test aid, needs no device: one strip's step of the measured balancing of the sweep's row segments:
 * n tiles with exclusive end rows ends[] (bottom to top, the last one = rows), the workgroups' measured durations[] (any unit),
 * the static weights[] of their places, rows_with_ice[rows] != 0 where a row holds anything to compute (or NULL: every row).
 * new_ends[]: every boundary half-way to where the running sum of the rows' cost reaches the tiles' shares; *total (may be
 * NULL): the strip's cost, 0 if the durations give nothing to go by (new_ends = ends then).  -2: bad arguments. */
int cice_debug_balance_strip(int rows, int n, const int32_t *ends, const double *durations, const double *weights,
                             const unsigned char *rows_with_ice, int32_t *new_ends, double *total);
/* evp -> transport WITHOUT a PCIe round trip.  In step_dynamics `call evp(dt)` is followed at once by `call
 * transport_remap(dt)` (source/ice_step_mod.F90:575-584): uvel, vvel come up from the device and go straight down again,
 * aicen, vicen are uploaded twice, and the rest of the state waits for the link while it idles during the subcycle loop.
 * cice_transport_chain(ctx, fields) names the host arrays the transport calls of this context will be given; from then on
 *   - every cice_evp call also STARTS the upload of aice0, trcrn, vsnon, eicen, esnon into the transport's buffers (on the
 *     copy streams, behind its own inputs: they travel while the subcycle loop runs), and
 *   - the next cice_transport_remap that is given the chained arrays uploads nothing: those five are there, uvel, vvel,
 *     aicen, vicen are taken from the dynamics on the device.
 * CONTRACT (the caller's statement, as for cice_evp_adopt_thermo_state): between the START of a cice_evp call and the
 * transport call that follows it, nothing writes to the seven state arrays, and uvel, vvel, aicen, vicen are the arrays
 * cice_evp was given -- true for step_dynamics of every driver under drivers/.  A transport call that does not follow a
 * cice_evp call (or is given other arrays) uploads as usual: the library enforces "follows": only a cice_evp call that
 * RETURNED CICE_OK arms the chain, and every other entry that touches or implies a newer host state (cice_evp_upload /
 * _prepare / _subcycles / _finish / _step, cice_evp_adopt_thermo_state, every cice_thermo_* / cice_step_therm1* entry,
 * cice_transport_upwind, a failed cice_transport_remap) disarms it.  fields = NULL ends the chain.  The Fortran drop-in
 * sets it up on its FIRST transport_remap call (the first step's evp is therefore unchained) when CICE4_AMD_CHAIN=1 is in
 * the environment. */
int cice_transport_chain(cice_ctx *ctx, const cice_transport_fields *fields);
/* advection = 'upwind' (source/ice_transport_driver.F90:672-834 transport_upwind with state_to_work :1570, upwind_field
 * :1796, work_to_state :1686 and compute_tracers, source/ice_itd.F90:1482): first-order donor-cell transport of aice0, of
 * every category's area, volumes and tracers and of the layer enthalpies, then bound_state, on the device.  HTE, HTN,
 * tarea: (nx_block, ny_block, nblocks) of this rank; nt_Tsfc: 1-based index of the surface temperature among the tracers
 * (it becomes Tocnfrz where a category's area vanishes).  Same host arrays as cice_transport_remap, updated in place. */
int cice_transport_upwind_init(cice_ctx *ctx, const cice_transport_config *cfg, int nt_Tsfc, const double *HTE,
                               const double *HTN, const double *tarea);
int cice_transport_upwind(cice_ctx *ctx, double dt, const cice_transport_fields *f);
/* Test aid: make the next cice_transport_remap stop after kernel stage stop_stage (0: run through) and / or copy
 * work array `which` (-1: none) to `out`; *count = its length in doubles. */
int cice_transport_debug(cice_ctx *ctx, int stop_stage, int which, double *out, long long *count);

/* Which reference build this library replaces: "standalone" (libcice4_amd.so: drivers/cice4/ice_constants.F90, the
 * compile-time cosw = 1, sinw = 0, dragio, chio of source/ice_dyn_evp.F90:76-88 and ice_therm_vertical.F90:680) or
 * "auscom" (libcice4_amd_auscom.so: the reference compiled -DAusCOM -Dcoupled, bld/Macros.nci:56-57, with
 * drivers/access-om/ice_constants.F90: MOM's cp_ocn and reference salinity; the ocean turning angle rotates with the
 * hemisphere in evp_prep2 / stepu / evp_finish, ice_dyn_evp.F90:910-913,1402-1408,1524-1536).  The two are separate
 * shared libraries, as the two are separate builds of the reference; a process may load both. */
const char *cice_build_flavour(void);
/* "auscom" flavour only (CICE_EINVAL in the stand-alone one): the namelist variables that build reads in
 * ice_init.F90:97,156,258-264 -- cosw, sinw (ocean turning angle), dragio (ice-ocean drag) of ice_dyn_evp.F90:91-97 --
 * and use_ocnslope (cpl_parameters: the tilt term from ss_tltx/ss_tlty instead of the geostrophic currents,
 * ice_dyn_evp.F90:919-933).  Defaults: 1, 0, 0.00536, 0.  Takes effect for every later call on this device (the
 * context's stream is drained first); a call that changes nothing returns at once, so the drop-in makes it before
 * every evp(dt). */
int cice_set_auscom(cice_ctx *ctx, double cosw, double sinw, double dragio, int use_ocnslope);
/* "auscom" flavour only: chio, the basal heat-transfer coefficient of frzmlt_bottom_lateral
 * (ice_therm_vertical.F90:57-60,692-694; namelist, default 0.006 = the stand-alone build's constant). */
int cice_thermo_set_chio(cice_ctx *ctx, double chio);

#ifdef __cplusplus
}
#endif
#endif
