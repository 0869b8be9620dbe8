// Kernels compiled for the model's own shape at nepmi_model_load (JIT cores).
//
// The reference's NEP kernels take the model's hyper-parameters by value (ParaMB / ANN, src/force/nep.cu:488-659, 774-861) and
// run any n_max / basis_size / l_max / neuron count through one code path.  This engine's fast kernels are templates on
// Shape<n_r, k_r, n_a, k_a, n_L, types> -- loops unrolled, per-atom tables in registers, packed FP32 -- and libnepmi.so carries
// five instantiations (engine_impl.h: S_PbTeA ... S_BZO) next to the run-time-shape kernels, whose register arrays live in
// scratch memory (PbTe 1 M atoms: 13.1 ms per step against 1.2: profiles/r5c_bench.json, pbte_generic_shape).  For a model of
// another shape the library therefore compiles ITSELF once more -- the same sources, `-DNEPMI_JIT_CORE
// -DNEPMI_JIT_SHAPE=n_r,k_r,n_a,k_a,n_L,types`: that shape only, ~40 s of hipcc -- into a JIT core
//     libnepmi_jit_<shape>_<hash of the sources>.so
// kept in <directory of libnepmi.so>/jit/ (cores built ahead of time, e.g. by __graft_entry__.build()) or in the user's cache
// ($NEPMI_JIT_CACHE, else ~/.cache/nepmi), loads it (dlopen, RTLD_LOCAL) and lets it serve the model: every handle carries the
// function table of the library that made it (capi_dispatch.inc), so the caller keeps talking to libnepmi.so's C ABI.
//
// NEPMI_JIT=0 turns it off (the run-time-shape kernels serve the model); hipcc is $NEPMI_HIPCC, else /opt/rocm/bin/hipcc,
// else `hipcc` on PATH; the sources are $NEPMI_SRC_DIR, else <directory of libnepmi.so>/../csrc.  When no compiler or no
// sources are found, or the compilation fails, the model is served by the run-time-shape kernels and one line on stderr says
// so.  Several processes (one per GPU) asking for the same core: one compiles (lock file), the others wait for the file.
// Shapes outside the templates' reach stay with the run-time-shape kernels: l_max_3body != 4, the optional 4-body rows.
#pragma once
#include <dlfcn.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <fcntl.h>
#include <unistd.h>

#include <chrono>
#include <ctime>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>

struct nepmi_api;
extern "C" const nepmi_api nepmi_self_api;

namespace nepmi {
namespace jit {

inline bool file_exists(const std::string& p)
{
  struct stat st;
  return ::stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}

inline std::string lib_dir()
{
  Dl_info info;
  if (dladdr((const void*)&nepmi_self_api, &info) && info.dli_fname) {
    std::string p = info.dli_fname;
    const size_t k = p.rfind('/');
    return k == std::string::npos ? std::string(".") : p.substr(0, k);
  }
  return ".";
}

inline std::string src_dir()
{
  if (const char* e = std::getenv("NEPMI_SRC_DIR"))
    return e;
  return lib_dir() + "/../csrc";
}

inline const char* const* source_files()
{
  static const char* const files[] = {
    "engine.hip", "nep_model.cpp", "transport_tcp.cpp", "engine_impl.h", "capi_impl.h", "capi_jit.h", "capi_dispatch.inc",
    "dist_bodies.h", "dist_impl.h", "dist_capi_impl.h", "nep_dev.h", "nep_bodies.h", "nep_window.h", "nep_scatter.h",
    "nep_fused.h", "nep_brick.h", "nep_highl.h", "nep_highl_tables.h", "nep_invariants_extra.h", "nep_md.h", "nep_model.h", "tersoff_bodies.h",
    "../../include/nepmi.h", nullptr};
  return files;
}

// FNV-1a over the sources: a core is only ever loaded by a library built from the same text
inline bool source_hash(const std::string& dir, uint64_t& h)
{
  h = 1469598103934665603ull;
  for (const char* const* f = source_files(); *f; ++f) {
    std::ifstream in(dir + "/" + *f, std::ios::binary);
    if (!in)
      return false;
    char buf[1 << 16];
    while (in) {
      in.read(buf, sizeof buf);
      const std::streamsize n = in.gcount();
      for (std::streamsize i = 0; i < n; ++i) {
        h ^= (unsigned char)buf[i];
        h *= 1099511628211ull;
      }
    }
  }
  return true;
}

inline std::string hipcc_path()
{
  if (const char* e = std::getenv("NEPMI_HIPCC"))
    return e;
  if (file_exists("/opt/rocm/bin/hipcc"))
    return "/opt/rocm/bin/hipcc";
  return "hipcc";
}

inline std::string cache_dir()
{
  if (const char* e = std::getenv("NEPMI_JIT_CACHE"))
    return e;
  const char* home = std::getenv("HOME");
  return std::string(home ? home : "/tmp") + "/.cache/nepmi";
}

inline void mkdirs(const std::string& p)
{
  std::string cur;
  for (size_t i = 0; i <= p.size(); ++i) {
    if (i == p.size() || p[i] == '/') {
      if (!cur.empty())
        ::mkdir(cur.c_str(), 0755);
    }
    if (i < p.size())
      cur.push_back(p[i]);
  }
}

struct ShapeKey {
  int nr, kr, na, ka, nl, ts;
  std::string name() const
  {
    char b[96];
    std::snprintf(b, sizeof b, "%d_%d_%d_%d_%d_%d", nr, kr, na, ka, nl, ts);
    return b;
  }
  std::string macro() const
  {
    char b[96];
    std::snprintf(b, sizeof b, "%d,%d,%d,%d,%d,%d", nr, kr, na, ka, nl, ts);
    return b;
  }
};

// the file name of the core for `key` built from the sources in `src`, or "" when the sources cannot be read
inline std::string core_name(const ShapeKey& key, const std::string& src)
{
  uint64_t h = 0;
  if (!source_hash(src, h))
    return "";
  char hb[32];
  std::snprintf(hb, sizeof hb, "%016llx", (unsigned long long)h);
  return "libnepmi_jit_" + key.name() + "_" + hb + ".so";
}

// Compile the core into `dir` (created if need be).  Returns the path, or "" (with `why` set).
inline std::string build_core(const ShapeKey& key, const std::string& dir, std::string& why)
{
  const std::string src = src_dir();
  const std::string name = core_name(key, src);
  if (name.empty()) {
    why = "the kernel sources were not found in " + src + " (NEPMI_SRC_DIR)";
    return "";
  }
  mkdirs(dir);
  const std::string out = dir + "/" + name, lock = out + ".lock", log = out + ".log";
  if (file_exists(out))
    return out;
  int fd = ::open(lock.c_str(), O_CREAT | O_EXCL | O_WRONLY, 0644);
  if (fd < 0) { // a lock left behind by a process that died while compiling (older than a quarter of an hour): take it over
    struct stat st;
    if (::stat(lock.c_str(), &st) == 0 && std::time(nullptr) - st.st_mtime > 900) {
      ::unlink(lock.c_str());
      fd = ::open(lock.c_str(), O_CREAT | O_EXCL | O_WRONLY, 0644);
    }
  }
  if (fd < 0) {
    // another process compiles this core: wait for it (a stale lock of a killed process: give up after ten minutes)
    for (int i = 0; i < 1200; ++i) {
      if (file_exists(out))
        return out;
      if (!file_exists(lock))
        break;
      std::this_thread::sleep_for(std::chrono::milliseconds(500));
    }
    if (file_exists(out))
      return out;
    why = "timed out waiting for another process to compile " + out + " (remove " + lock + " if it is stale)";
    return "";
  }
  ::close(fd);
  char tmp[64];
  std::snprintf(tmp, sizeof tmp, ".tmp%d", (int)::getpid());
  const std::string cmd = hipcc_path() + " --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-pass-failed -Wl,-Bsymbolic -Wl,-rpath,/opt/rocm/lib" +
                          " -DNEPMI_JIT_CORE -DNEPMI_JIT_SHAPE=" + key.macro() + " -o '" + out + tmp + "' '" + src + "/engine.hip' '" + src +
                          "/nep_model.cpp' '" + src + "/transport_tcp.cpp' -ldl > '" + log + "' 2>&1";
  std::fprintf(stderr, "nepmi: compiling the NEP kernels for this model's shape (n_max %d %d, basis_size %d %d, %d invariant rows, %s): "
                       "one-off, about a minute, kept as %s\n",
               key.nr, key.na, key.kr, key.ka, key.nl, key.ts ? "type-pure lists" : "any number of types", out.c_str());
  const int rc = std::system(cmd.c_str());
  std::string result;
  if (rc == 0 && file_exists(out + tmp) && ::rename((out + tmp).c_str(), out.c_str()) == 0) {
    result = out;
    ::unlink(log.c_str());
  } else {
    why = "hipcc failed (log: " + log + ")";
    ::unlink((out + tmp).c_str());
  }
  ::unlink(lock.c_str());
  return result;
}

inline const nepmi_api* load_core(const std::string& path, std::string& why)
{
  void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!h) {
    why = std::string("dlopen: ") + dlerror();
    return nullptr;
  }
  typedef const nepmi_api* (*get_api)(void);
  get_api f = reinterpret_cast<get_api>(dlsym(h, "nepmi_core_api"));
  if (!f) {
    why = "no nepmi_core_api in " + path;
    return nullptr;
  }
  return f();
}

// the core that serves models of this shape: loaded once per process; nullptr = the run-time-shape kernels of this library
inline const nepmi_api* core_for(const ShapeKey& key)
{
  static std::mutex mu;
  static std::map<std::string, const nepmi_api*> cores;
  std::lock_guard<std::mutex> lk(mu);
  auto it = cores.find(key.name());
  if (it != cores.end())
    return it->second;
  const nepmi_api* api = nullptr;
  std::string why;
  const std::string name = core_name(key, src_dir());
  std::string path;
  if (!name.empty()) {
    for (const std::string& d : {lib_dir() + "/jit", cache_dir()})
      if (path.empty() && file_exists(d + "/" + name))
        path = d + "/" + name;
  }
  const char* mode = std::getenv("NEPMI_JIT");
  const bool may_build = !(mode && mode[0] == '2'); // NEPMI_JIT=2: cores that exist already, never the compiler
  if (path.empty() && may_build)
    path = build_core(key, cache_dir(), why);
  if (!path.empty())
    api = load_core(path, why);
  if (!api)
    std::fprintf(stderr, "nepmi: no kernels compiled for this model's shape (%s): the run-time-shape kernels serve it, several "
                         "times slower\n", why.empty() ? "no core found, NEPMI_JIT=2" : why.c_str());
  if (api || may_build) // (a failed compilation is not tried again by this process; a look-up without the compiler may be)
    cores[key.name()] = api;
  return api;
}

} // namespace jit
} // namespace nepmi
