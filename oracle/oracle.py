"""TEST INFRASTRUCTURE ONLY -- ctypes loader for the plain-C CPU restatement
(oracle/_build/libcice_oracle.so, built by `make -C oracle`).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Same array conventions and call shapes as oracle/refapi.py."""
import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libcice_oracle.so")
NCAT, NILYR, NSLYR, MAX_NTRCR = 5, 4, 1, 5


def build():
    subprocess.check_call(["make", "-s", "-C", HERE])


def _p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


class EvpParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("dtei", "dte2T", "denom1", "denom2", "rcon", "ecci")] + \
               [("ndte", C.c_int), ("evp_damping", C.c_int)]


class ThermoCfg(C.Structure):
    _fields_ = [("salin", C.c_double * (NILYR + 1)), ("Tmlt", C.c_double * (NILYR + 1)),
                ("ustar_min", C.c_double), ("l_brine", C.c_int), ("heat_capacity", C.c_int),
                ("calc_Tsfc", C.c_int), ("conduct", C.c_int), ("tr_iage", C.c_int),
                ("nt_Tsfc", C.c_int), ("nt_iage", C.c_int)]


_DOM_PTRS = ("dxt", "dyt", "dxhy", "dyhx", "cxp", "cyp", "cxm", "cym", "tarea", "uarea", "tarear",
             "uarear", "tinyarea", "fcor")


class Domain(C.Structure):
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("nblocks", C.c_int),
                ("ilo", C.c_void_p), ("ihi", C.c_void_p), ("jlo", C.c_void_p), ("jhi", C.c_void_p),
                ("ncopy", C.c_int), ("hsrc", C.c_void_p), ("hdst", C.c_void_p),
                ("nfill", C.c_int), ("hfill", C.c_void_p)] + \
               [(n, C.c_void_p) for n in _DOM_PTRS] + \
               [("tmask", C.c_void_p), ("umask", C.c_void_p), ("kstrength", C.c_int),
                ("krdg_partic", C.c_int), ("krdg_redist", C.c_int), ("mu_rdg", C.c_double),
                ("perturb_strength_ulp", C.c_int)]


_ST_IN = ("aice", "vice", "vsno", "aice0", "aicen", "vicen", "strairxT", "strairyT", "uocn", "vocn",
          "ss_tltx", "ss_tlty")
_ST_IO = ("fm", "strtltx", "strtlty", "strocnx", "strocny", "strintx", "strinty")
_ST_OUT = ("strairx", "strairy", "strength", "divu", "shear", "rdg_conv", "rdg_shear", "prs_sig",
           "strocnxT", "strocnyT")
SIG_NAMES = ("stressp_1", "stressp_2", "stressp_3", "stressp_4", "stressm_1", "stressm_2",
             "stressm_3", "stressm_4", "stress12_1", "stress12_2", "stress12_3", "stress12_4")


class EvpState(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _ST_IN] + \
               [("uvel", C.c_void_p), ("vvel", C.c_void_p), ("sig", C.c_void_p * 12),
                ("iceumask", C.c_void_p)] + \
               [(n, C.c_void_p) for n in _ST_IO] + [(n, C.c_void_p) for n in _ST_OUT] + \
               [("aiu", C.c_void_p), ("umass", C.c_void_p), ("icetmask", C.c_void_p)]


class Oracle:
    def __init__(self, omp=False, aus=False):
        """aus=True: the build with the access-om driver's constants, switched to the AusCOM/coupled branches
        (set_auscom / set_chio hold their namelist values).  omp=True: the -fopenmp build (the two EVP loops spread over OMP_NUM_THREADS host cores; same
        results) -- bench.py's all-cores cpu_baseline leg only."""
        path = LIB.replace("libcice_oracle.so", "libcice_oracle_omp.so") if omp else LIB
        if aus:
            path = LIB.replace("libcice_oracle.so", "libcice_oracle_aus.so")
        if not os.path.exists(path):
            build()
        self.lib = C.CDLL(path)
        self.lib.orc_thermo_vertical.restype = C.c_int
        self.lib.orc_evp_subcycles_only.restype = C.c_double
        self.p = EvpParams()
        self.tc = ThermoCfg()
        self.strength_params = (1, 1, 1, 4.0)
        self.aus = aus
        if aus:
            self.set_auscom(True)
            self.set_chio()

    def set_evp_parameters(self, dt, ndte, damping=False):
        self.lib.orc_set_evp_parameters(C.c_double(dt), C.c_int(ndte), C.c_int(int(damping)),
                                        C.byref(self.p))
        return {n: getattr(self.p, n) for n in ("dtei", "dte2T", "denom1", "denom2", "rcon", "ecci")}

    def set_strength_parameters(self, kstrength=1, krdg_partic=1, krdg_redist=1, mu_rdg=4.0):
        self.strength_params = (kstrength, krdg_partic, krdg_redist, mu_rdg)

    @staticmethod
    def _sigptr(sig):
        arr = (C.c_void_p * 12)()
        for k in range(12):
            arr[k] = sig[k].ctypes.data
        return arr

    def stress(self, ksub, icellt, indxti, indxtj, uvel, vvel, g, strength, sig, diag, str8):
        ny, nx = uvel.shape
        self.lib.orc_stress(C.byref(self.p), C.c_int(nx), C.c_int(ny), C.c_int(ksub),
                            C.c_int(icellt), _p(indxti), _p(indxtj), _p(uvel), _p(vvel),
                            _p(g["dxt"]), _p(g["dyt"]), _p(g["dxhy"]), _p(g["dyhx"]), _p(g["cxp"]),
                            _p(g["cyp"]), _p(g["cxm"]), _p(g["cym"]), _p(g["tarear"]),
                            _p(g["tinyarea"]), _p(strength), self._sigptr(sig), _p(diag["shear"]),
                            _p(diag["divu"]), _p(diag["prs_sig"]), _p(diag["rdg_conv"]),
                            _p(diag["rdg_shear"]), _p(str8))

    def set_auscom(self, on, cosw=1.0, sinw=0.0, dragio=0.00536, use_ocnslope=False):
        """the -DAusCOM -Dcoupled build's variants of evp_prep2 / stepu / evp_finish (module-wide state of the library)"""
        self.lib.orc_set_auscom(C.c_int(int(on)), C.c_double(cosw), C.c_double(sinw), C.c_double(dragio),
                                C.c_int(int(use_ocnslope)))

    def set_chio(self, chio=0.006):
        self.lib.orc_set_chio(C.c_double(chio))

    def stepu(self, icellu, indxui, indxuj, aiu, str8, uocn, vocn, waterx, watery, forcex, forcey,
              umassdtei, fm, uarear, strocnx, strocny, strintx, strinty, uvel, vvel):
        ny, nx = uvel.shape
        self.lib.orc_stepu(C.c_int(nx), C.c_int(ny), C.c_int(icellu), _p(indxui), _p(indxuj),
                           _p(aiu), _p(str8), _p(uocn), _p(vocn), _p(waterx), _p(watery),
                           _p(forcex), _p(forcey), _p(umassdtei), _p(fm), _p(uarear), _p(strocnx),
                           _p(strocny), _p(strintx), _p(strinty), _p(uvel), _p(vvel))

    def evp_prep1(self, ilo, ihi, jlo, jhi, aice, vice, vsno, tmask, strairxT, strairyT):
        ny, nx = aice.shape
        strairx = np.zeros((ny, nx)); strairy = np.zeros((ny, nx)); tmass = np.zeros((ny, nx))
        icetmask = np.zeros((ny, nx), np.int32)
        tm = np.ascontiguousarray(tmask, np.int32)
        self.lib.orc_evp_prep1(C.c_int(nx), C.c_int(ny), C.c_int(ilo), C.c_int(ihi), C.c_int(jlo),
                               C.c_int(jhi), _p(aice), _p(vice), _p(vsno), _p(tm), _p(strairxT),
                               _p(strairyT), _p(strairx), _p(strairy), _p(tmass), _p(icetmask))
        return strairx, strairy, tmass, icetmask

    def evp_prep2(self, ilo, ihi, jlo, jhi, a):
        ny, nx = a["aiu"].shape
        icellt = C.c_int(0); icellu = C.c_int(0)
        lists = [np.zeros(nx * ny, np.int32) for _ in range(4)]
        self.lib.orc_evp_prep2(C.byref(self.p), C.c_int(nx), C.c_int(ny), C.c_int(ilo), C.c_int(ihi),
                               C.c_int(jlo), C.c_int(jhi), C.byref(icellt), C.byref(icellu),
                               *[_p(l) for l in lists], _p(a["aiu"]), _p(a["umass"]),
                               _p(a["umassdtei"]), _p(a["fcor"]), _p(a["umask"]), _p(a["uocn"]),
                               _p(a["vocn"]), _p(a["strairx"]), _p(a["strairy"]), _p(a["ss_tltx"]),
                               _p(a["ss_tlty"]), _p(a["icetmask"]), _p(a["iceumask"]), _p(a["fm"]),
                               _p(a["strtltx"]), _p(a["strtlty"]), _p(a["strocnx"]),
                               _p(a["strocny"]), _p(a["strintx"]), _p(a["strinty"]),
                               _p(a["waterx"]), _p(a["watery"]), _p(a["forcex"]), _p(a["forcey"]),
                               self._sigptr(a["sig"]), _p(a["uvel"]), _p(a["vvel"]))
        return icellt.value, icellu.value, lists

    def evp_finish(self, icellu, indxui, indxuj, uvel, vvel, uocn, vocn, aiu, strocnx, strocny,
                   strocnxT, strocnyT):
        ny, nx = uvel.shape
        self.lib.orc_evp_finish(C.c_int(nx), C.c_int(ny), C.c_int(icellu), _p(indxui), _p(indxuj),
                                _p(uvel), _p(vvel), _p(uocn), _p(vocn), _p(aiu), _p(strocnx),
                                _p(strocny), _p(strocnxT), _p(strocnyT))

    def evp_finish_fm(self, icellu, indxui, indxuj, uvel, vvel, uocn, vocn, aiu, fm, strocnx, strocny,
                      strocnxT, strocnyT):
        ny, nx = uvel.shape
        self.lib.orc_evp_finish_fm(C.c_int(nx), C.c_int(ny), C.c_int(icellu), _p(indxui), _p(indxuj),
                                   _p(uvel), _p(vvel), _p(uocn), _p(vocn), _p(aiu), _p(fm), _p(strocnx),
                                   _p(strocny), _p(strocnxT), _p(strocnyT))

    def ice_strength(self, ilo, ihi, jlo, jhi, icells, indxi, indxj, aice, vice, aice0, aicen, vicen):
        ny, nx = aice.shape
        strength = np.zeros((ny, nx))
        ks, kp, kr, mu = self.strength_params
        self.lib.orc_ice_strength(C.c_int(ks), C.c_int(kp), C.c_int(kr), C.c_double(mu), C.c_int(nx),
                                  C.c_int(ny), C.c_int(ilo), C.c_int(ihi), C.c_int(jlo), C.c_int(jhi),
                                  C.c_int(icells), _p(indxi), _p(indxj), _p(aice), _p(vice),
                                  _p(aice0), _p(aicen), _p(vicen), _p(strength))
        return strength

    # ---- whole evp ------------------------------------------------------
    def make_domain(self, dom, grid, perturb_strength_ulp=0):
        """dom: dict(nx, ny, nblocks, ilo.., hsrc, hdst[, hfill]); grid: dict of (nb,ny,nx)."""
        d = Domain()
        d.nx, d.ny, d.nblocks = dom["nx"], dom["ny"], dom["nblocks"]
        keep = []
        for n in ("ilo", "ihi", "jlo", "jhi"):
            a = np.ascontiguousarray(dom[n], np.int32); keep.append(a)
            setattr(d, n, a.ctypes.data)
        hs = np.ascontiguousarray(dom["hsrc"], np.int32); hd = np.ascontiguousarray(dom["hdst"], np.int32)
        hf = np.ascontiguousarray(dom.get("hfill", np.zeros(0)), np.int32)
        keep += [hs, hd, hf]
        d.ncopy, d.hsrc, d.hdst = len(hs), hs.ctypes.data, hd.ctypes.data
        d.nfill, d.hfill = len(hf), hf.ctypes.data
        for n in _DOM_PTRS:
            a = np.ascontiguousarray(grid[n], np.float64); keep.append(a)
            setattr(d, n, a.ctypes.data)
        for n in ("tmask", "umask"):
            a = np.ascontiguousarray(grid[n], np.int32); keep.append(a)
            setattr(d, n, a.ctypes.data)
        d.kstrength, d.krdg_partic, d.krdg_redist, d.mu_rdg = self.strength_params
        d.perturb_strength_ulp = perturb_strength_ulp
        d._keep = keep
        return d

    def make_state(self, s):
        """s: dict of arrays named as the reference's module arrays; modified in place."""
        st = EvpState()
        for n in _ST_IN + ("uvel", "vvel", "iceumask") + _ST_IO + _ST_OUT:
            setattr(st, n, s[n].ctypes.data)
        for k, n in enumerate(SIG_NAMES):
            st.sig[k] = s[n].ctypes.data
        for n in ("aiu", "umass", "icetmask"):
            setattr(st, n, s[n].ctypes.data if n in s else None)
        return st

    def evp(self, d, s):
        st = self.make_state(s)
        self.lib.orc_evp(C.byref(d), C.byref(self.p), C.byref(st))

    def evp_subcycles_only(self, d, s, nsub):
        st = self.make_state(s)
        return self.lib.orc_evp_subcycles_only(C.byref(d), C.byref(self.p), C.byref(st), C.c_int(nsub))

    # ---- thermo -----------------------------------------------------------
    def init_thermo(self, heat_capacity=True, calc_Tsfc=True, conduct="MU71", ustar_min=0.05):
        self.lib.orc_init_thermo(C.c_int(int(heat_capacity)), C.c_int(int(calc_Tsfc)),
                                 C.c_int(0 if conduct == "MU71" else 1), C.c_double(ustar_min),
                                 C.byref(self.tc))
        return np.array(self.tc.salin[:]), np.array(self.tc.Tmlt[:])

    THERMO_ARGS = ("aicen", "trcrn", "vicen", "vsnon", "eicen", "esnon", "flw", "potT", "Qa",
                   "rhoa", "fsnow", "fbot", "Tbot", "lhcoef", "shcoef", "fswsfc", "fswint",
                   "fswthrun", "Sswabs", "Iswabs", "fsurfn", "fcondtopn", "fsensn", "flatn",
                   "fswabsn", "flwoutn", "evapn", "freshn", "fsaltn", "fhocnn", "meltt", "melts",
                   "meltb", "congel", "snoice", "mlt_onset", "frz_onset")

    def thermo_vertical(self, dt, icells, indxi, indxj, a, yday=1.0):
        ny, nx = a["aicen"].shape
        istop = C.c_int(0); jstop = C.c_int(0)
        ls = self.lib.orc_thermo_vertical(C.byref(self.tc), C.c_int(nx), C.c_int(ny), C.c_double(dt),
                                          C.c_int(icells), _p(indxi), _p(indxj),
                                          *[_p(a[k]) for k in self.THERMO_ARGS], C.c_double(yday),
                                          C.byref(istop), C.byref(jstop))
        return ls, istop.value, jstop.value

    def frzmlt_bottom_lateral(self, ilo, ihi, jlo, jhi, dt, aice, frzmlt, eicen, esnon, sst, Tf,
                              strocnxT, strocnyT):
        ny, nx = aice.shape
        Tbot = np.zeros((ny, nx)); fbot = np.zeros((ny, nx)); rside = np.zeros((ny, nx))
        self.lib.orc_frzmlt_bottom_lateral(C.byref(self.tc), C.c_int(nx), C.c_int(ny), C.c_int(ilo),
                                           C.c_int(ihi), C.c_int(jlo), C.c_int(jhi), C.c_double(dt),
                                           _p(aice), _p(frzmlt), _p(eicen), _p(esnon), _p(sst),
                                           _p(Tf), _p(strocnxT), _p(strocnyT), _p(Tbot), _p(fbot),
                                           _p(rside))
        return Tbot, fbot, rside

    MERGE_ORDER = ("strairx", "strairy", "fsurf", "fcondtop", "fsens", "flat", "fswabs", "flwout", "evap",
                   "Tref", "Qref", "fresh", "fsalt", "fhocn", "fswthru", "meltt", "meltb", "melts", "congel",
                   "snoice")

    def merge_fluxes(self, icells, indxi, indxj, aicen, flw, catn, acc):
        """catn / acc: dicts keyed by MERGE_ORDER ((ny,nx) arrays); acc modified in place."""
        ny, nx = aicen.shape
        cp = (C.c_void_p * 20)(*[catn[k].ctypes.data for k in self.MERGE_ORDER])
        ap = (C.c_void_p * 20)(*[acc[k].ctypes.data for k in self.MERGE_ORDER])
        self.lib.orc_merge_fluxes(C.c_int(nx), C.c_int(ny), C.c_int(icells), _p(indxi), _p(indxj), _p(aicen),
                                  _p(flw), cp, ap)
