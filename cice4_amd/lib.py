"""ctypes binding of libcice4_amd.so (C-ABI in include/cice4_amd.h).

This is the Python stand-in for the Fortran ISO_C_BINDING shim
(cice4_amd/fortran/): same entry points, same argument order.  There is NO
CPU fallback: if the HIP library is missing or a device call fails, the call
raises.  Arrays are C-contiguous numpy arrays whose memory is the reference's
column-major layout: Fortran (nx,ny[,k][,nblocks]) <-> numpy ([nblocks,][k,] ny, nx).
"""
import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIBPATH = os.path.join(HERE, "libcice4_amd.so")
NCAT, NILYR, NSLYR, MAX_NTRCR = 5, 4, 1, 5

SIG_NAMES = ("stressp_1", "stressp_2", "stressp_3", "stressp_4", "stressm_1", "stressm_2",
             "stressm_3", "stressm_4", "stress12_1", "stress12_2", "stress12_3", "stress12_4")
EVP_GRID = ("dxt", "dyt", "dxhy", "dyhx", "cxp", "cyp", "cxm", "cym", "tarea", "uarea", "tarear",
            "uarear", "tinyarea", "fcor", "tmask", "umask")
EVP_GRID_OPT = ("HTN", "HTE")
EVP_IN = ("aice", "vice", "vsno", "aice0", "aicen", "vicen", "strairxT", "strairyT", "uocn", "vocn",
          "ss_tltx", "ss_tlty")
EVP_IO = ("uvel", "vvel") + SIG_NAMES + ("iceumask", "fm", "strtltx", "strtlty", "strocnx", "strocny",
                                          "strintx", "strinty")
EVP_OUT = ("strairx", "strairy", "strength", "divu", "shear", "rdg_conv", "rdg_shear", "prs_sig",
           "strocnxT", "strocnyT")
THERMO_STATE = ("aicen", "trcrn", "vicen", "vsnon", "eicen", "esnon")
THERMO_FORCING = ("flw", "potT", "Qa", "rhoa", "fsnow", "fbot", "Tbot")
THERMO_CAT_IN = ("lhcoef", "shcoef")
THERMO_SW = ("fswsfc", "fswint", "fswthrun", "Sswabs", "Iswabs")
THERMO_OUT = ("fsurfn", "fcondtopn", "fsensn", "flatn", "fswabsn", "flwoutn", "evapn", "freshn",
              "fsaltn", "fhocnn", "meltt", "melts", "meltb", "congel", "snoice")
THERMO_ONSET = ("mlt_onset", "frz_onset")
THERMO_ARGS = THERMO_STATE + ("flw", "potT", "Qa", "rhoa", "fsnow", "fbot", "Tbot", "lhcoef",
                              "shcoef") + THERMO_SW + THERMO_OUT + THERMO_ONSET


class CiceError(RuntimeError):
    pass


class EvpGrid(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in EVP_GRID + EVP_GRID_OPT]


class EvpConfig(C.Structure):
    _fields_ = [("ndte", C.c_int), ("evp_damping", C.c_int), ("kstrength", C.c_int),
                ("krdg_partic", C.c_int), ("krdg_redist", C.c_int), ("mu_rdg", C.c_double)]


class EvpFields(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in EVP_IN + EVP_IO + EVP_OUT]


class ThermoConfig(C.Structure):
    _fields_ = [("heat_capacity", C.c_int), ("calc_Tsfc", C.c_int), ("conduct", C.c_int),
                ("ustar_min", C.c_double), ("tr_iage", C.c_int), ("nt_Tsfc", C.c_int),
                ("nt_iage", C.c_int)]


MERGE_ORDER = ("strairxT", "strairyT", "fsurf", "fcondtop", "fsens", "flat", "fswabs", "flwout", "evap", "Tref",
               "Qref", "fresh", "fsalt", "fhocn", "fswthru", "meltt", "meltb", "melts", "congel", "snoice")


class MergeFields(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("aicen_init", "strairxn", "strairyn", "Trefn", "Qrefn")] + \
               [("acc", C.c_void_p * 20)]


class TransportConfig(C.Structure):
    _fields_ = [("ntrcr", C.c_int), ("trcr_depend", C.c_int * 5)]


class TransportGrid(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("HTN", "HTE", "dxt", "dyt", "dxu", "dyu", "tarear", "hm")]


class TransportFields(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("aice0", "aicen", "trcrn", "vicen", "vsnon", "eicen", "esnon", "uvel", "vvel")]


class FrzmltFields(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("aice", "frzmlt", "sst", "Tf", "strocnxT", "strocnyT", "Tbot", "fbot", "rside")]


class AtmoFields(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("uatm", "vatm", "wind", "zlvl", "strax", "stray")] + \
               [("calc_strair", C.c_int)] + \
               [(n, C.c_void_p) for n in ("strairxn", "strairyn", "Trefn", "Qrefn", "lhcoef", "shcoef")]


class ThermoFields(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in THERMO_STATE + THERMO_FORCING + THERMO_CAT_IN + THERMO_SW
                + THERMO_OUT + THERMO_ONSET]


_libs = {}


def load(flavour="standalone"):
    """Load the HIP library; raises if it has not been built (python -c 'import
    __graft_entry__ as g; g.build()' or make -C cice4_amd/csrc).  flavour "auscom": libcice4_amd_auscom.so, the
    build that replaces a reference compiled -DAusCOM -Dcoupled (include/cice4_amd.h: cice_build_flavour)."""
    if flavour not in _libs:
        if flavour not in ("standalone", "auscom"):
            raise CiceError(f"unknown flavour {flavour!r}")
        path = LIBPATH if flavour == "standalone" else LIBPATH.replace(".so", "_auscom.so")
        if not os.path.exists(path):
            raise CiceError(f"{path} not found: build the HIP extension first (make -C cice4_amd/csrc)")
        lib = C.CDLL(path)
        lib.cice_last_error.restype = C.c_char_p
        lib.cice_last_error.argtypes = [C.c_void_p]
        lib.cice_build_flavour.restype = C.c_char_p
        if lib.cice_build_flavour().decode() != flavour:
            raise CiceError(f"{path} reports flavour {lib.cice_build_flavour().decode()!r}")
        _libs[flavour] = lib
    return _libs[flavour]


def _p(a, dtype=None):
    if a is None:
        return None
    if dtype is not None and a.dtype != dtype:
        raise TypeError(f"expected {dtype}, got {a.dtype}")
    if not a.flags["C_CONTIGUOUS"]:
        raise ValueError("array must be C-contiguous")
    return a.ctypes.data_as(C.c_void_p)


def _f8(a):
    return _p(a, np.float64)


def _i4(a):
    return _p(a, np.int32)


class Context:
    """One rank = one GPU (cice_ctx)."""

    def __init__(self, device=-1, flavour="standalone"):
        self.lib = load(flavour)
        self.flavour = flavour
        self.h = C.c_void_p()
        rc = self.lib.cice_create(C.byref(self.h), C.c_int(device))
        if rc:
            raise CiceError("cice_create failed")
        self.nx = self.ny = self.nblocks = 0
        self._keep = []

    def close(self):
        if self.h:
            self.lib.cice_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc:
            msg = self.lib.cice_last_error(self.h)
            raise CiceError(f"cice4_amd error {rc}: {msg.decode() if msg else ''}")

    def sync(self):
        self._ck(self.lib.cice_device_sync(self.h))

    def set_auscom(self, cosw=1.0, sinw=0.0, dragio=0.00536, use_ocnslope=False):
        """flavour "auscom" only: the namelist variables of the coupled build's dynamics (cice_set_auscom)"""
        self._ck(self.lib.cice_set_auscom(self.h, C.c_double(cosw), C.c_double(sinw), C.c_double(dragio),
                                          C.c_int(int(use_ocnslope))))

    def set_chio(self, chio=0.006):
        """flavour "auscom" only: basal heat-transfer coefficient of frzmlt_bottom_lateral (cice_thermo_set_chio)"""
        self._ck(self.lib.cice_thermo_set_chio(self.h, C.c_double(chio)))

    def diag_stream_copy(self, n_doubles):
        ms = C.c_float(0.0)
        self._ck(self.lib.cice_diag_stream_copy(self.h, C.c_longlong(n_doubles), C.byref(ms)))
        return ms.value

    # ---- domain -------------------------------------------------------
    def domain_create(self, nxg, nyg, bsx, bsy, ew=1, ns=0, rank=0, npx=1, npy=1):
        self._ck(self.lib.cice_domain_create(self.h, nxg, nyg, bsx, bsy, ew, ns, rank, npx, npy))
        info = (C.c_int * 9)()
        self._ck(self.lib.cice_domain_info(self.h, info))
        self.nx, self.ny, self.nblocks = info[0], info[1], info[2]
        self.dinfo = dict(nx=info[0], ny=info[1], nblocks=info[2], nblocks_tot=info[3], ncopy=info[4],
                          nsend=info[5], nrecv=info[6], nsend_elems=info[7], nrecv_elems=info[8],
                          nxg=nxg, nyg=nyg)
        return self.domain()

    def domain_create_map(self, nxg, nyg, bsx, bsy, owner, ew=1, ns=0, rank=0, nranks=1, local_id=None):
        """Any block->task map; owner[g] = rank or -1 (eliminated land block)."""
        owner = np.ascontiguousarray(owner, np.int32)
        lid = None if local_id is None else np.ascontiguousarray(local_id, np.int32)
        self._ck(self.lib.cice_domain_create_map(self.h, nxg, nyg, bsx, bsy, ew, ns, rank, nranks, _i4(owner),
                                                 None if lid is None else _i4(lid)))
        info = (C.c_int * 9)()
        self._ck(self.lib.cice_domain_info(self.h, info))
        self.nx, self.ny, self.nblocks = info[0], info[1], info[2]
        self.dinfo = dict(nx=info[0], ny=info[1], nblocks=info[2], nblocks_tot=info[3], ncopy=info[4],
                          nsend=info[5], nrecv=info[6], nsend_elems=info[7], nrecv_elems=info[8],
                          nxg=nxg, nyg=nyg)
        return self.domain()

    def domain_list(self, name, loc=1):
        n = C.c_int(0)
        self._ck(self.lib.cice_domain_list(self.h, name.encode(), loc, C.byref(n), None))
        out = np.zeros(n.value, np.int32)
        if n.value:
            self._ck(self.lib.cice_domain_list(self.h, name.encode(), loc, C.byref(n), _i4(out)))
        return out

    def domain_create_slabs(self, nxg, nyg, nblocks_y, ew=1, ns=0, rank=0, nranks=1, overlap=0):
        """j-slabs of full width, optionally extended by `overlap` rows (wide halo)."""
        self._ck(self.lib.cice_domain_create_slabs(self.h, nxg, nyg, nblocks_y, ew, ns, rank, nranks, overlap))
        info = (C.c_int * 9)()
        self._ck(self.lib.cice_domain_info(self.h, info))
        self.nx, self.ny, self.nblocks = info[0], info[1], info[2]
        self.dinfo = dict(nx=info[0], ny=info[1], nblocks=info[2], nblocks_tot=info[3], ncopy=info[4],
                          nsend=info[5], nrecv=info[6], nsend_elems=info[7], nrecv_elems=info[8],
                          nxg=nxg, nyg=nyg, overlap=overlap)
        return self.domain()

    def domain(self):
        """dict describing the local blocks + on-rank halo list (host logic, no GPU needed)."""
        d = dict(self.dinfo)
        cols = {k: [] for k in ("ilo", "ihi", "jlo", "jhi", "i0", "j0", "gid", "owner", "own_jlo", "own_jhi")}
        for b in range(self.nblocks):
            info = (C.c_int * 10)()
            self._ck(self.lib.cice_domain_block(self.h, b, info))
            for k, v in zip(cols, info):
                cols[k].append(v)
        d.update({k: np.array(v, np.int32) for k, v in cols.items()})
        src = np.zeros(d["ncopy"], np.int32); dst = np.zeros(d["ncopy"], np.int32)
        if d["ncopy"]:
            self._ck(self.lib.cice_domain_halo_local(self.h, _i4(src), _i4(dst)))
        d["hsrc"], d["hdst"] = src, dst
        nr = C.c_int(0)
        self._ck(self.lib.cice_domain_halo_refresh(self.h, C.byref(nr), None, None))
        rs = np.zeros(nr.value, np.int32); rd = np.zeros(nr.value, np.int32)
        if nr.value:
            self._ck(self.lib.cice_domain_halo_refresh(self.h, C.byref(nr), _i4(rs), _i4(rd)))
        d["rsrc"], d["rdst"] = rs, rd
        d["hfill"] = self.domain_list("hfill")
        return d

    def apply_halo_lists(self, a, loc=1, kind=1, fill=0):
        """The halo update the device performs, in numpy on a host array (nlev?, nblocks, ny, nx) -- for CPU tests of
        the lists: on-rank copies, fill of ghost cells facing eliminated blocks, tripole fold (single-rank domains)."""
        d = self.domain()
        n = self.nblocks * self.ny * self.nx
        v = a.reshape(-1, n)
        v[:, d["hdst"]] = v[:, d["hsrc"]]
        if len(d["hfill"]):
            v[:, d["hfill"]] = fill
        lsrc, bidx = self.domain_list("fold_lsrc"), self.domain_list("fold_bidx")
        if len(lsrc):
            sgn = 1 if kind == 1 else -1
            rows = int(bidx.max()) // self.dinfo["nxg"] + 1     # 2: fold through U points, 3: through T points
            buf = np.full((v.shape[0], rows * self.dinfo["nxg"]), fill, a.dtype)
            buf[:, bidx] = v[:, lsrc]
            lo, hi = self.domain_list("fold_lo", loc), self.domain_list("fold_hi", loc)
            if len(lo):
                if a.dtype == np.int32:   # nint(0.5_dbl_kind*(x1 + isign*x2)), halves away from zero
                    t = 0.5 * (buf[:, lo] + sgn * buf[:, hi]).astype(np.float64)
                    x = (np.sign(t) * np.floor(np.abs(t) + 0.5)).astype(np.int32)
                else:
                    x = (a.dtype.type(0.5) * (buf[:, lo] + a.dtype.type(sgn) * buf[:, hi])).astype(a.dtype)
                buf[:, lo] = x
                buf[:, hi] = a.dtype.type(sgn) * x
            dst, src = self.domain_list("fold_dst", loc), self.domain_list("fold_src", loc)
            v[:, dst] = a.dtype.type(sgn) * buf[:, src]
        return a

    def halo_msgs(self, direction):
        """direction 0 / 1: ghost-cell messages sent / received; 2 / 3: tripole top rows sent / received (receive
        addresses are indices into the global fold buffer).  The list ends at the first CICE_EINVAL for a message
        index; a context without a domain or a bad direction is an error, not an empty list, and for directions
        0 / 1 the count is checked against cice_domain_info."""
        if direction not in (0, 1, 2, 3):
            raise CiceError(f"halo_msgs: direction {direction} not in 0..3")
        if not getattr(self, "dinfo", None):
            raise CiceError("halo_msgs: no domain (call domain_create* first)")
        out = []
        m = -1
        while True:
            m += 1
            peer = C.c_int(); cnt = C.c_int()
            rc = self.lib.cice_domain_halo_msg(self.h, direction, m, C.byref(peer), C.byref(cnt), None)
            if rc == -1:        # CICE_EINVAL: past the last message (context and direction were checked above)
                break
            self._ck(rc)
            addr = np.zeros(cnt.value, np.int32)
            self._ck(self.lib.cice_domain_halo_msg(self.h, direction, m, None, None, _i4(addr)))
            out.append((peer.value, addr))
        if direction in (0, 1):
            want = self.dinfo["nsend" if direction == 0 else "nrecv"]
            if len(out) != want:
                raise CiceError(f"halo_msgs({direction}): {len(out)} messages listed, cice_domain_info says {want}")
        return out

    # ---- communication --------------------------------------------------
    def comm_unique_id(self):
        buf = C.create_string_buffer(128)
        rc = self.lib.cice_comm_unique_id(buf)
        if rc:
            raise CiceError(f"cice_comm_unique_id failed ({rc})")
        return buf.raw

    def comm_init(self, uid, rank, nranks):
        self._ck(self.lib.cice_comm_init(self.h, C.c_char_p(uid), rank, nranks))

    def check_sizes(self, ncat, nilyr, nslyr, max_ntrcr):
        """CiceError unless the host model's ice_domain_size parameters are the library's compile-time sizes"""
        self._ck(self.lib.cice_check_sizes(self.h, ncat, nilyr, nslyr, max_ntrcr))

    def comm_init_local(self, link_id, rank, nranks):
        """in-process link instead of an RCCL communicator: the ranks are contexts of this process (tests)"""
        self._ck(self.lib.cice_comm_init_local(self.h, link_id, rank, nranks))

    def comm_init_shm(self, name, rank, nranks, box_bytes=64 << 20):
        """the same between processes of one host (a file under /dev/shm)"""
        self._ck(self.lib.cice_comm_init_shm(self.h, name.encode(), rank, nranks, C.c_longlong(box_bytes)))

    def comm_init_mirror(self, rank, nranks):
        """timing aid: this context is rank `rank` of `nranks`, alone on its device; its messages come back to it"""
        self._ck(self.lib.cice_comm_init_mirror(self.h, rank, nranks))

    def comm_count(self):
        """ranks of this context's communicator as RCCL counts them (0 before comm_init)"""
        n = C.c_int(0)
        self._ck(self.lib.cice_comm_count(self.h, C.byref(n)))
        return n.value

    # ---- EVP -----------------------------------------------------------------
    def evp_init(self, grid, ndte=120, evp_damping=False, kstrength=1, krdg_partic=1, krdg_redist=1,
                 mu_rdg=4.0):
        g = EvpGrid()
        for n in EVP_GRID:
            a = grid[n]
            setattr(g, n, (_i4(a) if n in ("tmask", "umask") else _f8(a)))
        for n in EVP_GRID_OPT:
            setattr(g, n, _f8(grid[n]) if n in grid else None)
        cfg = EvpConfig(ndte, int(evp_damping), kstrength, krdg_partic, krdg_redist, mu_rdg)
        self._ck(self.lib.cice_evp_init(self.h, C.byref(cfg), C.byref(g)))

    @staticmethod
    def _evp_fields(s, need_out=True):
        f = EvpFields()
        for n in EVP_IN + EVP_IO:
            if s.get(n) is None:          # left to the device (evp_adopt_thermo_state)
                setattr(f, n, None)
                continue
            setattr(f, n, (_i4(s[n]) if n == "iceumask" else _f8(s[n])))
        for n in EVP_OUT:
            setattr(f, n, _f8(s[n]) if (n in s) else None)
        return f

    def evp(self, dt, s):
        """Drop-in evp(dt): s holds the module arrays (modified in place)."""
        f = self._evp_fields(s)
        self._ck(self.lib.cice_evp(self.h, C.c_double(dt), C.byref(f)))

    def evp_adopt_thermo_state(self):
        """aicen, vicen (+ the aggregates) of the dynamics from the batched thermo state on the device; the next
        evp / evp_upload may leave aice, vice, vsno, aice0, aicen, vicen out (None)"""
        self._ck(self.lib.cice_evp_adopt_thermo_state(self.h))

    def evp_pin_fields(self, s):
        """Page-lock the arrays of s (they must stay alive and keep their addresses)."""
        f = self._evp_fields(s)
        self._ck(self.lib.cice_evp_pin_fields(self.h, C.byref(f)))

    def evp_upload(self, s):
        f = self._evp_fields(s)
        self._ck(self.lib.cice_evp_upload(self.h, C.byref(f)))

    def evp_download(self, s):
        f = self._evp_fields(s)
        self._ck(self.lib.cice_evp_download(self.h, C.byref(f)))

    def evp_download_stresses(self, s):
        f = self._evp_fields(s)
        self._ck(self.lib.cice_evp_download_stresses(self.h, C.byref(f)))

    def evp_step(self, dt):
        self._ck(self.lib.cice_evp_step(self.h, C.c_double(dt)))

    def evp_prepare(self, dt):
        self._ck(self.lib.cice_evp_prepare(self.h, C.c_double(dt)))

    def evp_subcycles(self, ksub0, nsub, timed=False):
        ms = C.c_float(0.0)
        self._ck(self.lib.cice_evp_subcycles(self.h, ksub0, nsub, C.byref(ms) if timed else None))
        return ms.value

    def evp_finish(self):
        self._ck(self.lib.cice_evp_finish(self.h))

    def evp_set_option(self, key, value):
        self._ck(self.lib.cice_evp_set_option(self.h, key.encode(), int(value)))

    def evp_get_info(self, key):
        v = C.c_int(0)
        self._ck(self.lib.cice_evp_get_info(self.h, key.encode(), C.byref(v)))
        return v.value

    def evp_peer_export(self):
        """(xu0, xu1, rprog, plane): device pointers of this rank's exchange copies / progress words and its plane size"""
        bufs = (C.c_void_p * 3)(); plane = C.c_longlong(0)
        self._ck(self.lib.cice_evp_peer_export(self.h, bufs, C.byref(plane)))
        return bufs[0], bufs[1], bufs[2], plane.value

    def evp_peer_connect(self, side, peer):
        """side 0: `peer` (an evp_peer_export tuple) is the rank to the south, 1: to the north"""
        self._ck(self.lib.cice_evp_peer_connect(self.h, side, C.c_void_p(peer[0]), C.c_void_p(peer[1]), C.c_void_p(peer[2]),
                                                C.c_longlong(peer[3])))

    def evp_peer_export_ipc(self):
        """(3 x 64-byte IPC handles, plane): the same buffers for a neighbour in another process"""
        h = C.create_string_buffer(192); plane = C.c_longlong(0)
        self._ck(self.lib.cice_evp_peer_export_ipc(self.h, h, C.byref(plane)))
        return h.raw, plane.value

    def evp_peer_connect_ipc(self, side, peer):
        self._ck(self.lib.cice_evp_peer_connect_ipc(self.h, side, C.c_char_p(peer[0]), C.c_longlong(peer[1])))

    def evp_peer_ranks(self):
        """the ranks this rank's block exchanges ghost cells with (ascending; any cartesian layout, one block per rank)"""
        n = C.c_int(0); r = (C.c_int32 * 8)()
        self._ck(self.lib.cice_evp_peer_ranks(self.h, C.byref(n), r))
        return [int(r[k]) for k in range(n.value)]

    def evp_peer_connect_rank(self, rank, peer):
        """`peer` (an evp_peer_export tuple) are the buffers of neighbouring rank `rank`"""
        self._ck(self.lib.cice_evp_peer_connect_rank(self.h, rank, C.c_void_p(peer[0]), C.c_void_p(peer[1]), C.c_void_p(peer[2]),
                                                     C.c_longlong(peer[3])))

    def evp_peer_connect_rank_ipc(self, rank, peer):
        self._ck(self.lib.cice_evp_peer_connect_rank_ipc(self.h, rank, C.c_char_p(peer[0]), C.c_longlong(peer[1])))

    def evp_debug(self, what):
        n = C.c_longlong(0)
        self._ck(self.lib.cice_evp_debug(self.h, what.encode(), None, C.byref(n)))
        out = np.zeros(n.value, np.int64)
        if n.value:
            self._ck(self.lib.cice_evp_debug(self.h, what.encode(), out.ctypes.data_as(C.c_void_p), C.byref(n)))
        return out

    def evp_active_cells(self):
        nt = C.c_longlong(); nu = C.c_longlong()
        self._ck(self.lib.cice_evp_active_cells(self.h, C.byref(nt), C.byref(nu)))
        return nt.value, nu.value

    def evp_stress(self, dt, ndte, damping, ksub, icellt, indxti, indxtj, uvel, vvel, g, strength,
                   sig, diag, str8):
        ny, nx = uvel.shape
        self._ck(self.lib.cice_evp_stress(
            self.h, C.c_double(dt), ndte, int(damping), nx, ny, ksub, icellt, _i4(indxti), _i4(indxtj),
            _f8(uvel), _f8(vvel), _f8(g["dxt"]), _f8(g["dyt"]), _f8(g["dxhy"]), _f8(g["dyhx"]),
            _f8(g["cxp"]), _f8(g["cyp"]), _f8(g["cxm"]), _f8(g["cym"]), _f8(g["tarear"]),
            _f8(g["tinyarea"]), _f8(strength), *[_f8(sig[k]) for k in range(12)], _f8(diag["shear"]),
            _f8(diag["divu"]), _f8(diag["prs_sig"]), _f8(diag["rdg_conv"]), _f8(diag["rdg_shear"]),
            _f8(str8)))

    def evp_stepu(self, icellu, indxui, indxuj, aiu, str8, uocn, vocn, waterx, watery, forcex, forcey,
                  umassdtei, fm, uarear, strocnx, strocny, strintx, strinty, uvel, vvel):
        ny, nx = uvel.shape
        self._ck(self.lib.cice_evp_stepu(
            self.h, nx, ny, icellu, _i4(indxui), _i4(indxuj), _f8(aiu), _f8(str8), _f8(uocn), _f8(vocn),
            _f8(waterx), _f8(watery), _f8(forcex), _f8(forcey), _f8(umassdtei), _f8(fm), _f8(uarear),
            _f8(strocnx), _f8(strocny), _f8(strintx), _f8(strinty), _f8(uvel), _f8(vvel)))

    def halo_update(self, field, loc=1, kind=1, fill=0):
        """field: (nlev?, nblocks, ny, nx) float64 / float32 / int32, updated in place.  loc / kind: field location
        and type codes of ice_HaloUpdate (tripole fold); fill: value for ghost cells facing eliminated blocks."""
        n = self.nblocks * self.ny * self.nx
        nlev = field.size // n
        assert field.flags["C_CONTIGUOUS"]
        p = field.ctypes.data_as(C.c_void_p)
        if field.dtype == np.float64:
            self._ck(self.lib.cice_halo_update_ex_r8(self.h, p, nlev, loc, kind, C.c_double(fill)))
        elif field.dtype == np.float32:
            self._ck(self.lib.cice_halo_update_ex_r4(self.h, p, nlev, loc, kind, C.c_float(fill)))
        else:
            assert field.dtype == np.int32
            self._ck(self.lib.cice_halo_update_ex_i4(self.h, p, nlev, loc, kind, C.c_int32(int(fill))))

    def halo_update_blocked(self, field, loc=1, kind=1, fill=0):
        """The same for a field in the reference's own array layout (nblocks, nz, ny, nx) -- block outermost, what the
        Fortran boundary module hands over (cice_halo_update_blocked_*): only the frame of each block travels."""
        nz = field.size // (self.nblocks * self.ny * self.nx)
        assert field.flags["C_CONTIGUOUS"] and field.shape[0] == self.nblocks
        p = field.ctypes.data_as(C.c_void_p)
        if field.dtype == np.float64:
            self._ck(self.lib.cice_halo_update_blocked_r8(self.h, p, nz, loc, kind, C.c_double(fill)))
        elif field.dtype == np.float32:
            self._ck(self.lib.cice_halo_update_blocked_r4(self.h, p, nz, loc, kind, C.c_float(fill)))
        else:
            assert field.dtype == np.int32
            self._ck(self.lib.cice_halo_update_blocked_i4(self.h, p, nz, loc, kind, C.c_int32(int(fill))))

    def halo_update_resident(self, field):
        """The same on a field kept in device memory: upload once, update through cice_halo_update_dev_r8/_i4
        (all levels in one message per neighbour, no staging, no allocation per call), download."""
        n = self.nblocks * self.ny * self.nx
        nlev = field.size // n
        dev = C.c_void_p()
        self._ck(self.lib.cice_device_alloc(self.h, C.c_size_t(field.nbytes), C.byref(dev)))
        try:
            self._ck(self.lib.cice_device_copy(self.h, dev, _p(field), C.c_size_t(field.nbytes), 1))
            fn = self.lib.cice_halo_update_dev_r8 if field.dtype == np.float64 else self.lib.cice_halo_update_dev_i4
            self._ck(fn(self.h, dev, nlev))
            self._ck(self.lib.cice_device_copy(self.h, _p(field), dev, C.c_size_t(field.nbytes), 0))
        finally:
            self._ck(self.lib.cice_device_free(self.h, dev))

    # ---- thermodynamics ----------------------------------------------------------
    def thermo_init(self, heat_capacity=True, calc_Tsfc=True, conduct="MU71", ustar_min=0.05,
                    tr_iage=True, nt_Tsfc=1, nt_iage=2):
        cfg = ThermoConfig(int(heat_capacity), int(calc_Tsfc), 0 if conduct == "MU71" else 1, ustar_min,
                           int(tr_iage), nt_Tsfc, nt_iage)
        salin = np.zeros(NILYR + 1); tmlt = np.zeros(NILYR + 1)
        self._ck(self.lib.cice_thermo_init(self.h, C.byref(cfg), _f8(salin), _f8(tmlt)))
        return salin, tmlt

    def thermo_vertical(self, dt, icells, indxi, indxj, a, yday=1.0):
        ny, nx = a["aicen"].shape
        ls = C.c_int32(0); istop = C.c_int32(0); jstop = C.c_int32(0)
        self._ck(self.lib.cice_thermo_vertical(
            self.h, nx, ny, C.c_double(dt), icells, _i4(indxi), _i4(indxj),
            *[_f8(a[k]) for k in THERMO_ARGS], C.c_double(yday), C.byref(ls), C.byref(istop),
            C.byref(jstop)))
        return ls.value, istop.value, jstop.value

    def thermo_set_option(self, key, value):
        self._ck(self.lib.cice_thermo_set_option(self.h, key.encode(), int(value)))

    def thermo_batch_alloc(self, nx, ny, nblocks):
        self._ck(self.lib.cice_thermo_batch_alloc(self.h, nx, ny, nblocks))

    @staticmethod
    def _thermo_fields(a, outputs=True):
        f = ThermoFields()
        for n, _t in ThermoFields._fields_:
            setattr(f, n, _f8(a[n]) if n in a else None)
        return f

    def host_register(self, arr):
        """Page-lock a numpy array that will be passed to evp()/thermo entries repeatedly."""
        self._ck(self.lib.cice_host_register(self.h, arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.nbytes)))

    def host_unregister_all(self):
        """Must precede the release of any array handed to host_register / evp_pin_fields."""
        self._ck(self.lib.cice_host_unregister_all(self.h))

    def thermo_batch_upload(self, a):
        f = self._thermo_fields(a)
        self._ck(self.lib.cice_thermo_batch_upload(self.h, C.byref(f)))

    def thermo_batch_step(self, dt, yday=1.0, timed=False):
        nupd = C.c_longlong(0); ms = C.c_float(0.0)
        st = [C.c_int32(0) for _ in range(5)]
        self._ck(self.lib.cice_thermo_batch_step(self.h, C.c_double(dt), C.c_double(yday), C.byref(nupd),
                                                 *[C.byref(x) for x in st],
                                                 C.byref(ms) if timed else None))
        return dict(n_updates=nupd.value, l_stop=st[0].value, istop=st[1].value, jstop=st[2].value,
                    nstop=st[3].value, bstop=st[4].value, ms=ms.value)

    def thermo_batch_download(self, a):
        f = self._thermo_fields(a)
        self._ck(self.lib.cice_thermo_batch_download(self.h, C.byref(f)))

    def thermo_batch_merge(self, percat, acc):
        """percat: aicen_init, strairxn, strairyn, Trefn, Qrefn ((nb,ncat,ny,nx)); acc: dict keyed by
        MERGE_ORDER ((nb,ny,nx)), updated in place (merge_fluxes for every category)."""
        f = MergeFields()
        for n in ("aicen_init", "strairxn", "strairyn", "Trefn", "Qrefn"):
            setattr(f, n, _f8(percat[n]))
        for k, n in enumerate(MERGE_ORDER):
            f.acc[k] = acc[n].ctypes.data
            assert acc[n].dtype == np.float64 and acc[n].flags["C_CONTIGUOUS"]
        self._ck(self.lib.cice_thermo_batch_merge(self.h, C.byref(f)))

    # ---- horizontal transport --------------------------------------------------
    def transport_init(self, grid, ntrcr=2, trcr_depend=(0, 1)):
        """grid: HTN, HTE, dxt, dyt, dxu, dyu, tarear, hm as (nblocks, ny, nx) arrays; tracer set as in ice_init.F90:848."""
        cfg = TransportConfig()
        cfg.ntrcr = ntrcr
        for k, d in enumerate(trcr_depend):
            cfg.trcr_depend[k] = d
        g = TransportGrid()
        self._tgrid = {n: np.ascontiguousarray(grid[n], np.float64) for n, _t in TransportGrid._fields_}
        for n in self._tgrid:
            setattr(g, n, _f8(self._tgrid[n]))
        self._ck(self.lib.cice_transport_init(self.h, C.byref(cfg), C.byref(g)))

    def transport_upwind_init(self, HTE, HTN, tarea, ntrcr=2, trcr_depend=(0, 1), nt_Tsfc=1):
        cfg = TransportConfig()
        cfg.ntrcr = ntrcr
        for k, d in enumerate(trcr_depend):
            cfg.trcr_depend[k] = d
        self._ck(self.lib.cice_transport_upwind_init(self.h, C.byref(cfg), C.c_int(nt_Tsfc), _f8(HTE), _f8(HTN), _f8(tarea)))

    def transport_upwind(self, dt, s):
        """advection = 'upwind' on the arrays of transport_remap, updated in place"""
        f = TransportFields()
        for n, _t in TransportFields._fields_:
            setattr(f, n, _f8(s[n]))
        self._ck(self.lib.cice_transport_upwind(self.h, C.c_double(dt), C.byref(f)))

    def transport_remap(self, dt, s):
        """s: aice0, uvel, vvel (nb,ny,nx); aicen, vicen, vsnon (nb,ncat,ny,nx); trcrn (nb,ncat,5,ny,nx); eicen
        (nb,ncat*nilyr,ny,nx); esnon (nb,ncat*nslyr,ny,nx); state updated in place.  Returns (l_stop, istop, jstop)."""
        f = TransportFields()
        for n, _t in TransportFields._fields_:
            setattr(f, n, _f8(s[n]))
        st = [C.c_int32(0) for _ in range(3)]
        self._ck(self.lib.cice_transport_remap(self.h, C.c_double(dt), C.byref(f), *[C.byref(x) for x in st]))
        return tuple(x.value for x in st)

    def transport_chain(self, s):
        """cice_transport_chain: s = the arrays the following transport_remap calls will be given (None ends the chain).
        Every evp(dt, ...) then prefetches the transport's state while it subcycles; a transport_remap that follows it with
        these arrays (and the uvel, vvel, aicen, vicen ARRAYS evp was given) uploads nothing."""
        if s is None:
            self._ck(self.lib.cice_transport_chain(self.h, None))
            self._chain = None
            return
        f = TransportFields()
        for n, _t in TransportFields._fields_:
            setattr(f, n, _f8(s[n]))
        self._chain = s          # keep the arrays alive: the library remembers their addresses
        self._ck(self.lib.cice_transport_chain(self.h, C.byref(f)))

    def transport_debug(self, stop_stage=0, which=-1):
        cnt = C.c_longlong(0)
        self._ck(self.lib.cice_transport_debug(self.h, stop_stage, which, None, C.byref(cnt)))
        if which < 0:
            return None
        out = np.zeros(cnt.value)
        self._ck(self.lib.cice_transport_debug(self.h, stop_stage, which, _f8(out), C.byref(cnt)))
        return out

    ATMO_OUT = ("strx", "stry", "Tref", "Qref", "delt", "delq", "lhcoef", "shcoef")

    def atmo_boundary_layer(self, sfctype, icells, indxi, indxj, a, calc_strair=True, strx=None, stry=None):
        """cice_atmo_boundary_layer (source/ice_atmo.F90:56).  a: Tsf, potT, uatm, vatm, wind, zlvl, Qa, rhoa (ny, nx);
        returns the eight outputs (strx, stry keep the given values when calc_strair is false)."""
        ny, nx = a["Tsf"].shape
        o = {k: np.zeros((ny, nx)) for k in self.ATMO_OUT}
        if strx is not None:
            o["strx"][...] = strx; o["stry"][...] = stry
        ins = [np.ascontiguousarray(a[k], np.float64) for k in ("Tsf", "potT", "uatm", "vatm", "wind", "zlvl", "Qa", "rhoa")]
        self._ck(self.lib.cice_atmo_boundary_layer(
            self.h, nx, ny, 0 if sfctype == "ice" else 1, icells, _i4(indxi), _i4(indxj), *[_f8(x) for x in ins],
            int(calc_strair), *[_f8(o[k]) for k in self.ATMO_OUT]))
        return o

    def step_therm1(self, dt, yday, state, fz, percat, acc, atm=None):
        """cice_step_therm1: one upload, frzmlt_bottom_lateral + thermo_vertical for every category + merge_fluxes on
        the device, one download.  state: thermo_batch_upload's dict (fbot/Tbot not needed); fz: aice, frzmlt, sst,
        Tf, strocnxT, strocnyT (+ optional outputs Tbot, fbot, rside), (nb,ny,nx); percat / acc as thermo_batch_merge
        (percat['aicen_init'] optional).  atm (cice_step_therm1_abl): uatm, vatm, wind, zlvl (+ strax, stray and
        calc_strair=False) -- atmo_boundary_layer runs on the device too, state's lhcoef/shcoef and percat's
        strairxn/strairyn/Trefn/Qrefn are not read; arrays found in atm under those six names are filled."""
        f = self._thermo_fields(state)
        z = FrzmltFields()
        for n, _t in FrzmltFields._fields_:
            setattr(z, n, _f8(fz[n]) if n in fz else None)
        m = MergeFields()
        for n in ("aicen_init", "strairxn", "strairyn", "Trefn", "Qrefn"):
            setattr(m, n, _f8(percat[n]) if n in percat else None)
        for k, n in enumerate(MERGE_ORDER):
            m.acc[k] = acc[n].ctypes.data
            assert acc[n].dtype == np.float64 and acc[n].flags["C_CONTIGUOUS"]
        nupd = C.c_longlong(0)
        st = [C.c_int32(0) for _ in range(5)]
        if atm is None:
            self._ck(self.lib.cice_step_therm1(self.h, C.c_double(dt), C.c_double(yday), C.byref(f), C.byref(z),
                                               C.byref(m), C.byref(nupd), *[C.byref(x) for x in st]))
        else:
            af = AtmoFields()
            for n, t in AtmoFields._fields_:
                if n == "calc_strair":
                    af.calc_strair = int(atm.get("calc_strair", True))
                else:
                    setattr(af, n, _f8(atm[n]) if n in atm else None)
            self._ck(self.lib.cice_step_therm1_abl(self.h, C.c_double(dt), C.c_double(yday), C.byref(f), C.byref(z),
                                                   C.byref(m), C.byref(af), C.byref(nupd), *[C.byref(x) for x in st]))
        return dict(n_updates=nupd.value, l_stop=st[0].value, istop=st[1].value, jstop=st[2].value,
                    nstop=st[3].value, bstop=st[4].value)

    def frzmlt_bottom_lateral(self, ilo, ihi, jlo, jhi, dt, aice, frzmlt, eicen, esnon, sst, Tf,
                              strocnxT, strocnyT):
        ny, nx = aice.shape
        Tbot = np.zeros((ny, nx)); fbot = np.zeros((ny, nx)); rside = np.zeros((ny, nx))
        self._ck(self.lib.cice_frzmlt_bottom_lateral(
            self.h, nx, ny, ilo, ihi, jlo, jhi, C.c_double(dt), _f8(aice), _f8(frzmlt), _f8(eicen),
            _f8(esnon), _f8(sst), _f8(Tf), _f8(strocnxT), _f8(strocnyT), _f8(Tbot), _f8(fbot),
            _f8(rside)))
        return Tbot, fbot, rside
