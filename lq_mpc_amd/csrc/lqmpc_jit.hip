// lqmpc_jit.hip -- the 16-lane-row kernels for shapes without a prebuilt instantiation, compiled at run time.
//
// The reference's API takes any horizon and any model size (LQ_MPC_Controller(N, ...), utils_class.py:23, 62; LQ_RDP_Behavior loops
// N_min..N_max, utils_class.py:409-466), the fast kernels are templates on (n_x, n_u, N).  Instead of a table of instantiations
// (rounds 1-2: eleven shapes, everything else 650x slower on the generic kernel) the device headers are embedded in the library as
// text (gen_jit_src.py) and the kernel of a new (shape, mode) is compiled on first use with hiprtc -- about two seconds, once: code
// objects are kept in memory for the process and, when a cache directory is set (lqmpc_jit_cache_dir; the Python binding points it
// at the package), on disk across processes.  hiprtc is loaded with dlopen: a machine without it falls back to the generic kernel.
// The compile itself needs no GPU (lqmpc_jit_compile can pre-build code objects on a build machine).
#include "lqmpc_common.h"
#include "lqmpc_r16_body.h"        // (host side: the LDS-size arithmetic of R16 is checked against the templates below)
#include "lqmpc_bounds_chip.h"
#include "../../include/lqmpc.h"

#include <hip/hiprtc.h>
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "lqmpc_jit_src.inc"

namespace lqmpc {

// ---- which shapes the 16-lane-row algorithm serves (lqmpc_r16_body.h / lqmpc_r16_setup.h) ----
//   n_x <= 8 and n_u <= 4 (register matrices of the set-up), n = N n_u <= 32 with four instances per wavefront (the smaller side of
//   an active-set system has <= 16 unknowns: one per lane), 32 < n <= 48 with one instance per wavefront (<= 24 unknowns).
bool jit_r16_shape(int nx, int nu, int N, int *lpi)
{
    if (lpi) *lpi = 0;
    if (nx < 1 || nu < 1 || N < 1 || nx > SETUP_MAX_NX || nu > SETUP_MAX_NU) return false;
    const int n = N * nu;
    if (n > 48) return false;
    if (lpi) *lpi = n <= 32 ? 16 : 64;
    return true;
}

// R16<NX, NU, N, LPI, PACKED>::INST restated for run-time dimensions (doubles of LDS per instance)
static int r16_inst(int nx, int nu, int N, int lpi, bool packed)
{
    const int n = N * nu, rb = (n + lpi - 1) / lpi, cs = 4 * (((n + 1) / 2 + 3) / 4), ldw = n + 1;
    const int pk = packed ? n * (n + 1) / 2 : n * ldw, vec = lpi * rb;
    const int cn0 = 3 * nx * nx + nu * nu + nx * nu, cn1 = (nx > nu ? nx : nu) * (2 * nx + 2 * nu), cn = cn0 > cn1 ? cn0 : cn1;
    const int oR = 2 * pk, oL = oR + 3 * vec, oC = oL + cs / 2;
    const int setup = n * nx, end = oC + cn + (cn & 1), oD = end > setup ? end : setup;
    return oD + 2;
}
static_assert(R16<4, 2, 10, 16, true>::INST == 584 && R16<2, 1, 30, 16, true>::INST == 1052 && R16<4, 2, 20, 64, true>::INST == 1904,
              "r16_inst() below restates this arithmetic: keep the two in step");

struct Build { int lpi, occ, waves, inst; };
static Build r16_build(int nx, int nu, int N)
{
    Build b{};
    jit_r16_shape(nx, nu, N, &b.lpi);
    const int n = N * nu;
    b.occ = ((b.lpi == 16 && n > 10) || b.lpi == 64) ? 2 : 1;             // as R16Build in lqmpc_r16.hip
    b.inst = r16_inst(nx, nu, N, b.lpi, b.occ == 2);
    const long long lds = (long long)(64 / b.lpi) * b.inst * 8;
    b.waves = (lds * 8 <= 160 * 1024) ? 2 : 1;                            // two waves per SIMD only where their LDS fits
    return b;
}

// ---- hiprtc through dlopen ----
struct Rtc {
    void *lib = nullptr;
    bool tried = false;
    hiprtcResult (*create)(hiprtcProgram *, const char *, const char *, int, const char **, const char **) = nullptr;
    hiprtcResult (*compile)(hiprtcProgram, int, const char **) = nullptr;
    hiprtcResult (*log_size)(hiprtcProgram, size_t *) = nullptr;
    hiprtcResult (*log)(hiprtcProgram, char *) = nullptr;
    hiprtcResult (*code_size)(hiprtcProgram, size_t *) = nullptr;
    hiprtcResult (*code)(hiprtcProgram, char *) = nullptr;
    hiprtcResult (*destroy)(hiprtcProgram *) = nullptr;
    hiprtcResult (*version)(int *, int *) = nullptr;
    bool ok() const { return create && compile && log_size && log && code_size && code && destroy; }
};
static Rtc g_rtc;

static bool rtc_load()
{
    if (g_rtc.tried) return g_rtc.ok();
    g_rtc.tried = true;
    for (const char *name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
        g_rtc.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (g_rtc.lib) break;
    }
    if (!g_rtc.lib) return false;
#define LQMPC_SYM(field, sym) g_rtc.field = (decltype(g_rtc.field))dlsym(g_rtc.lib, sym)
    LQMPC_SYM(create, "hiprtcCreateProgram"); LQMPC_SYM(compile, "hiprtcCompileProgram");
    LQMPC_SYM(log_size, "hiprtcGetProgramLogSize"); LQMPC_SYM(log, "hiprtcGetProgramLog");
    LQMPC_SYM(code_size, "hiprtcGetCodeSize"); LQMPC_SYM(code, "hiprtcGetCode");
    LQMPC_SYM(destroy, "hiprtcDestroyProgram"); LQMPC_SYM(version, "hiprtcVersion");
#undef LQMPC_SYM
    return g_rtc.ok();
}

// ---- the cache: code objects per (shape, mode); loaded modules per (device, shape, mode) ----
typedef std::tuple<int, int, int, int> Key;                     // nx, nu, N, mode
struct Loaded { hipModule_t mod = nullptr; hipFunction_t fn = nullptr; };
static std::mutex g_mu;
static std::map<Key, std::vector<char>> g_code;
static std::map<Key, std::string> g_failed;                     // shapes whose compile failed: the log, no retry
static std::map<std::tuple<int, int, int, int, int>, Loaded> g_loaded;
static std::map<Key, std::string> g_names;
static std::string g_dir;

static unsigned long long fnv(const std::string &s, unsigned long long h = 1469598103934665603ull)
{
    for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
    return h;
}

constexpr int JIT_BOUNDS_SMALL = 100, JIT_BOUNDS_BIG = 101;      // cache keys beside the solver's modes

// BigT<NX, NU, N, LPI>::INST restated for run-time dimensions
static int bounds_big_inst(int nx, int nu, int N, int lpi)
{
    const int n = N * nu, rb = (n + lpi - 1) / lpi, vec = lpi * rb;
    return n * (n + 1) + 4 * vec + N * nx * nu + 2 + 2;
}
static_assert(BigT<4, 2, 10, 16>::INST == 20 * 21 + 4 * 32 + 80 + 4 && BigT<2, 1, 6, 16>::INST == 6 * 7 + 4 * 16 + 12 + 4, "bounds_big_inst() restates this");

static std::string program_text(int nx, int nu, int N, int mode)
{
    char buf[1024];
    if (mode == JIT_BOUNDS_SMALL) {
        snprintf(buf, sizeof buf,
                 "#include \"lqmpc_bounds_chip.h\"\nnamespace lqmpc {\nextern \"C\" __global__ void __launch_bounds__(64) lqmpc_jit_kernel(BoundsParams p)\n"
                 "{ bounds_small<%d, %d, %d>(p); }\n}\n", nx, nu, N);
        return buf;
    }
    if (mode == JIT_BOUNDS_BIG) {
        const int lpi = N * nu <= 32 ? 16 : 64;
        snprintf(buf, sizeof buf,
                 "#include \"lqmpc_bounds_chip.h\"\nnamespace lqmpc {\nextern \"C\" __global__ void __launch_bounds__(64) lqmpc_jit_kernel(BoundsParams p)\n"
                 "{ extern __shared__ double lds_dyn[]; bounds_big<%d, %d, %d, %d>(p, lds_dyn); }\n}\n", nx, nu, N, lpi);
        return buf;
    }
    if (mode == MODE_PROBE) {
        snprintf(buf, sizeof buf,
                 "#include \"lqmpc_probe.h\"\nnamespace lqmpc {\nextern \"C\" __global__ void __launch_bounds__(64) lqmpc_jit_kernel(KParams p)\n"
                 "{ probe_body<%d, %d, %d>(p); }\n}\n", nx, nu, N);
        return buf;
    }
    const Build b = r16_build(nx, nu, N);
    snprintf(buf, sizeof buf,
             "#include \"lqmpc_r16_body.h\"\nnamespace lqmpc {\nextern \"C\" __global__ void __launch_bounds__(64, %d) lqmpc_jit_kernel(KParams p)\n"
             "{\n    using C = R16<%d, %d, %d, %d, %s>;\n    __shared__ double lds_raw[C::IPW * C::INST];\n"
             "    r16_body<%d, %d, %d, %d, %d, %d>(p, lds_raw, (long long)blockIdx.x * C::IPW, p.Bsz);\n}\n}\n",
             b.waves, nx, nu, N, b.lpi, b.occ == 2 ? "true" : "false", nx, nu, N, mode, b.lpi, b.occ);
    return buf;
}

static std::string cache_path(const std::string &text)
{
    if (g_dir.empty()) return "";
    unsigned long long h = fnv(text);
    // (only the headers the program includes enter the key: an edit of the bounds kernels leaves the cached solver kernels valid)
    const bool bounds = text.find("lqmpc_bounds_chip.h") != std::string::npos, probe = text.find("lqmpc_probe.h") != std::string::npos;
    for (int i = 0; i < k_jit_count; ++i) {
        const std::string nm = k_jit_names[i];
        const bool used = nm == "lqmpc_common.h" || (probe ? nm == "lqmpc_probe.h"
                          : (nm == "lqmpc_wg_linalg.h" || nm == "lqmpc_r16_setup.h" ||
                             (bounds ? (nm == "lqmpc_bounds.h" || nm == "lqmpc_bounds_chip.h") : nm == "lqmpc_r16_body.h")));
        if (used) h = fnv(k_jit_srcs[i], h);
    }
    int maj = 0, min = 0;
    if (g_rtc.version) g_rtc.version(&maj, &min);
    h = fnv(std::to_string(maj) + "." + std::to_string(min) + " gfx950 O3", h);
    char name[64];
    snprintf(name, sizeof name, "/lqmpc_%016llx.hsaco", h);
    return g_dir + name;
}

// code object of (shape, mode): memory, then disk, then hiprtc.  Returns nullptr (and the reason in err) when it cannot be had.
static const std::vector<char> *get_code(int nx, int nu, int N, int mode, std::string &err)
{
    const Key k{nx, nu, N, mode};
    auto it = g_code.find(k);
    if (it != g_code.end()) return &it->second;
    auto f = g_failed.find(k);
    if (f != g_failed.end()) { err = f->second; return nullptr; }
    if (!rtc_load()) { err = g_failed[k] = "hiprtc is not available (dlopen libhiprtc.so failed)"; return nullptr; }
    const std::string text = program_text(nx, nu, N, mode), path = cache_path(text);
    if (!path.empty()) {
        if (FILE *fp = fopen(path.c_str(), "rb")) {
            std::vector<char> code;
            char buf[65536];
            size_t got;
            while ((got = fread(buf, 1, sizeof buf, fp)) > 0) code.insert(code.end(), buf, buf + got);
            fclose(fp);
            if (code.size() > 64) return &(g_code[k] = std::move(code));
        }
    }
    hiprtcProgram prog = nullptr;
    if (g_rtc.create(&prog, text.c_str(), "lqmpc_jit.hip", k_jit_count, (const char **)k_jit_srcs, (const char **)k_jit_names) != HIPRTC_SUCCESS) {
        err = g_failed[k] = "hiprtcCreateProgram failed";
        return nullptr;
    }
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-DLQMPC_JIT"};
    const hiprtcResult rc = g_rtc.compile(prog, 5, opts);
    if (rc != HIPRTC_SUCCESS) {
        size_t ls = 0;
        g_rtc.log_size(prog, &ls);
        std::string log(ls, '\0');
        if (ls) g_rtc.log(prog, &log[0]);
        g_rtc.destroy(&prog);
        err = g_failed[k] = "hiprtc compile failed: " + log.substr(0, 2000);
        return nullptr;
    }
    size_t cs = 0;
    g_rtc.code_size(prog, &cs);
    std::vector<char> code(cs);
    g_rtc.code(prog, code.data());
    g_rtc.destroy(&prog);
    if (!path.empty()) {                                          // (best effort; written under a temporary name, then renamed)
        const std::string tmp = path + ".tmp" + std::to_string((long long)getpid());
        if (FILE *fp = fopen(tmp.c_str(), "wb")) {
            const bool ok = fwrite(code.data(), 1, code.size(), fp) == code.size();
            fclose(fp);
            if (!ok || rename(tmp.c_str(), path.c_str()) != 0) remove(tmp.c_str());
        }
    }
    return &(g_code[k] = std::move(code));
}

static const char *mode_tag(int mode)
{
    switch (mode) { case MODE_SOLVE: return "solve"; case MODE_ROLLOUT: return "rollout"; case MODE_MAXVN: return "maxvn";
                    case MODE_PROBE: return "probe"; default: return "sweep"; }
}

// the kernel of (shape, mode) on `device`, ready to launch; nullptr (reason in err) when the shape is outside the domain or the
// compile / load failed
static const Loaded *get_kernel(int device, int nx, int nu, int N, int mode, std::string &err)
{
    std::lock_guard<std::mutex> lock(g_mu);
    const std::tuple<int, int, int, int, int> lk{device, nx, nu, N, mode};
    auto it = g_loaded.find(lk);
    if (it != g_loaded.end()) return &it->second;
    const std::vector<char> *code = get_code(nx, nu, N, mode, err);
    if (!code) return nullptr;
    Loaded l;
    hipError_t e = hipModuleLoadData(&l.mod, code->data());
    if (e == hipSuccess) e = hipModuleGetFunction(&l.fn, l.mod, "lqmpc_jit_kernel");
    if (e != hipSuccess) {
        err = g_failed[Key{nx, nu, N, mode}] = std::string("hipModuleLoadData: ") + hipGetErrorString(e);
        return nullptr;
    }
    if (mode < JIT_BOUNDS_SMALL && !g_names.count(Key{nx, nu, N, 0})) {
        char nm[96];
        const Build b = r16_build(nx, nu, N);
        snprintf(nm, sizeof nm, "lqmpc_r%d_jit_kernel<%d,%d,%d>", b.lpi, nx, nu, N);
        g_names[Key{nx, nu, N, 0}] = nm;
    }
    return &(g_loaded[lk] = l);
}

bool jit_available(int device, int nx, int nu, int N, int mode, std::string *why)
{
    std::string err;
    if (!jit_r16_shape(nx, nu, N, nullptr)) { if (why) *why = "outside the 16-lane-row domain (nx <= 8, nu <= 4, N nu <= 48)"; return false; }
    const bool ok = get_kernel(device, nx, nu, N, mode, err) != nullptr;
    if (!ok && why) *why = err;
    return ok;
}

// launch the run-time compiled kernel of p's shape and mode (MODE_PROBE: the difficulty probe); false when it cannot be had
bool launch_jit(int device, const KParams &p, hipStream_t stream, const char **name, std::string *why)
{
    std::string err;
    const Loaded *l = get_kernel(device, p.nx, p.nu, p.N, p.mode, err);
    if (!l) { if (why) *why = err; return false; }
    unsigned grid;
    if (p.mode == MODE_PROBE) grid = (unsigned)((p.Bsz + 63) / 64);
    else {
        const int ipw = 64 / r16_build(p.nx, p.nu, p.N).lpi;
        grid = (unsigned)((p.Bsz + ipw - 1) / ipw);
    }
    KParams arg = p;
    void *args[] = {&arg};
    const hipError_t e = hipModuleLaunchKernel(l->fn, grid, 1, 1, 64, 1, 1, 0, stream, args, nullptr);
    if (e != hipSuccess) { if (why) *why = std::string("hipModuleLaunchKernel: ") + hipGetErrorString(e); return false; }
    if (name) {
        std::lock_guard<std::mutex> lock(g_mu);
        *name = g_names[Key{p.nx, p.nu, p.N, 0}].c_str();
    }
    return true;
}

int jit_lanes(int nx, int nu, int N) { int l = 0; return jit_r16_shape(nx, nu, N, &l) ? l : 0; }

// the on-chip pair of bound-coefficient kernels (lqmpc_bounds_chip.h) of a shape without prebuilt ones: nx <= 8, nu <= 4, any N nu <= 128
bool jit_bounds_shape(int nx, int nu, int N)
{
    if (nx < 1 || nu < 1 || N < 1 || nx > SETUP_MAX_NX || nu > SETUP_MAX_NU || N * nu > 128) return false;
    const int lpi = N * nu <= 32 ? 16 : 64;
    return (long long)(64 / lpi) * bounds_big_inst(nx, nu, N, lpi) * 8 <= 160 * 1024;
}

bool launch_jit_bounds(int device, const BoundsParams &p, hipStream_t stream, std::string *why)
{
    std::string err;
    if (!jit_bounds_shape(p.nx, p.nu, p.N)) { if (why) *why = "outside the domain of the on-chip bounds kernels"; return false; }
    const Loaded *k1 = get_kernel(device, p.nx, p.nu, p.N, JIT_BOUNDS_SMALL, err);
    const Loaded *k2 = k1 ? get_kernel(device, p.nx, p.nu, p.N, JIT_BOUNDS_BIG, err) : nullptr;
    if (!k1 || !k2) { if (why) *why = err; return false; }
    const int lpi = p.N * p.nu <= 32 ? 16 : 64, ipw = 64 / lpi;
    const size_t lds = (size_t)ipw * bounds_big_inst(p.nx, p.nu, p.N, lpi) * 8;
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute((const void *)k2->fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { if (why) *why = std::string("hipFuncSetAttribute: ") + hipGetErrorString(e); return false; }
    }
    BoundsParams arg = p;
    void *args[] = {&arg};
    const long long m = p.b1 - p.b0;
    hipError_t e = hipModuleLaunchKernel(k1->fn, (unsigned)((m + 3) / 4), 1, 1, 64, 1, 1, 0, stream, args, nullptr);
    if (e == hipSuccess) e = hipModuleLaunchKernel(k2->fn, (unsigned)((m + ipw - 1) / ipw), 1, 1, 64, 1, 1, (unsigned)lds, stream, args, nullptr);
    if (e != hipSuccess) { if (why) *why = std::string("hipModuleLaunchKernel: ") + hipGetErrorString(e); return false; }
    return true;
}

}  // namespace lqmpc

extern "C" {

// Directory for compiled code objects (kept across processes); NULL or "" = memory only.  Process-wide.
int lqmpc_jit_cache_dir(const char *dir)
{
    std::lock_guard<std::mutex> lock(lqmpc::g_mu);
    lqmpc::g_dir = dir ? dir : "";
    if (!lqmpc::g_dir.empty()) {
        mkdir(lqmpc::g_dir.c_str(), 0777);                                 // (one level; an existing directory is fine)
        if (access(lqmpc::g_dir.c_str(), W_OK) != 0) { lqmpc::g_dir.clear(); return LQMPC_ERR_BAD_ARG; }
    }
    return 0;
}

// Compile (or find in the cache directory) the kernels of one shape without touching a GPU: every mode of the 16-lane-row kernel and
// the probe.  Returns the number of code objects now available, or a negative error (shape outside the domain, no hiprtc, compile error:
// the text goes to `log` if given).
int lqmpc_jit_compile(int nx, int nu, int N, char *log, int log_len)
{
    if (log && log_len > 0) log[0] = '\0';
    if (!lqmpc::jit_r16_shape(nx, nu, N, nullptr)) return LQMPC_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> lock(lqmpc::g_mu);
    int count = 0;
    for (int mode : {(int)lqmpc::MODE_SOLVE, (int)lqmpc::MODE_ROLLOUT, (int)lqmpc::MODE_MAXVN, (int)lqmpc::MODE_SWEEP, (int)lqmpc::MODE_PROBE}) {
        std::string err;
        if (lqmpc::get_code(nx, nu, N, mode, err)) ++count;
        else {
            if (log && log_len > 0) snprintf(log, (size_t)log_len, "%s: %s", lqmpc::mode_tag(mode), err.c_str());
            return LQMPC_ERR_UNSUPPORTED;
        }
    }
    return count;
}

// ... and the pair of on-chip bound-coefficient kernels of a shape (nx <= 8, nu <= 4, LDS image within 160 KiB); returns 2
int lqmpc_jit_compile_bounds(int nx, int nu, int N, char *log, int log_len)
{
    if (log && log_len > 0) log[0] = '\0';
    if (!lqmpc::jit_bounds_shape(nx, nu, N)) return LQMPC_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> lock(lqmpc::g_mu);
    for (int mode : {lqmpc::JIT_BOUNDS_SMALL, lqmpc::JIT_BOUNDS_BIG}) {
        std::string err;
        if (!lqmpc::get_code(nx, nu, N, mode, err)) {
            if (log && log_len > 0) snprintf(log, (size_t)log_len, "bounds kernel %d: %s", mode - lqmpc::JIT_BOUNDS_SMALL, err.c_str());
            return LQMPC_ERR_UNSUPPORTED;
        }
    }
    return 2;
}

}  // extern "C"
