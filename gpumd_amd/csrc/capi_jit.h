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
#include <vector>
#include <cerrno>
#include <sys/file.h>
#include <sys/wait.h>

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
    "nep_fused.h", "nep_highl.h", "nep_highl_tables.h", "nep_invariants_extra.h", "nep_md.h", "nep_model.h", "tersoff_bodies.h",
    "../../include/nepmi.h", nullptr};
  return files;
}

// FNV-1a over the sources.  The hash of the text THIS library was built from is baked in at build time (Makefile:
// -DNEPMI_SRC_HASH, tools/build_jit_core.py --hash): it names the cores this library may load, and a core reports the hash it
// was built with (nepmi_core_abi) -- a core and a library of different text can differ in the order of the function table or
// in the layout of the handles.  The files on disk are read only when a core has to be COMPILED, and only used when they
// still hash to the baked value.
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

#ifndef NEPMI_SRC_HASH
#define NEPMI_SRC_HASH 0ull // a build without the Makefile: no JIT cores (nothing to match them with)
#endif
inline uint64_t baked_hash() { return (uint64_t)NEPMI_SRC_HASH; }

inline std::string hipcc_path()
{
  if (const char* e = std::getenv("NEPMI_HIPCC"))
    return e;
  if (file_exists("/opt/rocm/bin/hipcc"))
    return "/opt/rocm/bin/hipcc";
  return "hipcc";
}

// $NEPMI_JIT_CACHE, else ~/.cache/nepmi; "" when neither is known (no HOME): a predictable path under a world-writable
// directory is not a place to dlopen shared objects from
inline std::string cache_dir()
{
  if (const char* e = std::getenv("NEPMI_JIT_CACHE"))
    return e;
  const char* home = std::getenv("HOME");
  if (!home || !home[0])
    return "";
  return std::string(home) + "/.cache/nepmi";
}

inline void mkdirs(const std::string& p)
{
  std::string cur;
  for (size_t i = 0; i <= p.size(); ++i) {
    if (i == p.size() || p[i] == '/') {
      if (!cur.empty())
        ::mkdir(cur.c_str(), 0700);
    }
    if (i < p.size())
      cur.push_back(p[i]);
  }
}

// the cache directory must be ours alone before anything in it is loaded into the process
inline bool private_dir(const std::string& p)
{
  struct stat st;
  return ::stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode) && st.st_uid == ::geteuid() && (st.st_mode & (S_IWGRP | S_IWOTH)) == 0;
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

// the file name of the core for `key` that matches this library
inline std::string core_name(const ShapeKey& key)
{
  char hb[32];
  std::snprintf(hb, sizeof hb, "%016llx", (unsigned long long)baked_hash());
  return "libnepmi_jit_" + key.name() + "_" + hb + ".so";
}

// hipcc with an argument vector (no shell: paths may hold any character), output into `log`
inline int run_compiler(const std::vector<std::string>& argv, const std::string& log)
{
  const pid_t pid = ::fork();
  if (pid < 0)
    return -1;
  if (pid == 0) {
    const int fd = ::open(log.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0600);
    if (fd >= 0) {
      ::dup2(fd, 1);
      ::dup2(fd, 2);
      ::close(fd);
    }
    std::vector<char*> av;
    for (const std::string& a : argv)
      av.push_back(const_cast<char*>(a.c_str()));
    av.push_back(nullptr);
    ::execvp(av[0], av.data());
    ::_exit(127);
  }
  int status = 0;
  while (::waitpid(pid, &status, 0) < 0 && errno == EINTR) {
  }
  return WIFEXITED(status) ? WEXITSTATUS(status) : -1;
}

// Compile the core into `dir` (created if need be).  Returns the path, or "" (with `why` set).  One process compiles, the
// others wait on the lock file (flock: released by the kernel when its holder dies, however it dies).
inline std::string build_core(const ShapeKey& key, const std::string& dir, std::string& why)
{
  if (dir.empty()) {
    why = "no cache directory (set NEPMI_JIT_CACHE or HOME)";
    return "";
  }
  const std::string src = src_dir();
  uint64_t on_disk = 0;
  if (!source_hash(src, on_disk)) {
    why = "the kernel sources were not found in " + src + " (NEPMI_SRC_DIR)";
    return "";
  }
  if (on_disk != baked_hash()) {
    why = "the sources in " + src + " are not the ones this library was built from (rebuild the library, or point NEPMI_SRC_DIR at its sources)";
    return "";
  }
  mkdirs(dir);
  if (!private_dir(dir)) {
    why = dir + " is not a directory owned by this user and writable by nobody else";
    return "";
  }
  const std::string name = core_name(key);
  const std::string out = dir + "/" + name, lock = out + ".lock";
  if (file_exists(out))
    return out;
  const int lfd = ::open(lock.c_str(), O_CREAT | O_RDWR, 0600);
  if (lfd < 0) {
    why = "cannot create " + lock;
    return "";
  }
  if (::flock(lfd, LOCK_EX) != 0) {
    ::close(lfd);
    why = "cannot lock " + lock;
    return "";
  }
  std::string result;
  if (file_exists(out)) { // another process compiled it while this one waited
    result = out;
  } else {
    char tag[64];
    std::snprintf(tag, sizeof tag, ".%d.%llx", (int)::getpid(),
                  (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count());
    const std::string tmp = out + tag + ".tmp", log = out + tag + ".log";
    char hash_def[64];
    std::snprintf(hash_def, sizeof hash_def, "-DNEPMI_SRC_HASH=0x%016llxull", (unsigned long long)baked_hash());
    const std::vector<std::string> argv = {hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-pass-failed",
                                           "-Wl,-Bsymbolic", "-Wl,-rpath,/opt/rocm/lib", "-DNEPMI_JIT_CORE", "-DNEPMI_JIT_SHAPE=" + key.macro(),
                                           hash_def, "-o", tmp, src + "/engine.hip", src + "/nep_model.cpp", src + "/transport_tcp.cpp", "-ldl"};
    std::fprintf(stderr, "nepmi: compiling the NEP kernels for this model's shape (n_max %d %d, basis_size %d %d, %d invariant rows, %s): "
                         "one-off, about a minute, kept as %s\n",
                 key.nr, key.na, key.kr, key.ka, key.nl, key.ts ? "type-pure lists" : "any number of types", out.c_str());
    const int rc = run_compiler(argv, log);
    if (rc == 0 && file_exists(tmp) && ::rename(tmp.c_str(), out.c_str()) == 0) {
      result = out;
      ::unlink(log.c_str());
    } else {
      why = rc == 127 ? "no compiler (" + hipcc_path() + ": NEPMI_HIPCC)" : "hipcc failed (log: " + log + ")";
      ::unlink(tmp.c_str());
    }
  }
  // (a waiter that opened the file before this unlink gets the lock of the orphaned inode next, finds the core in place and
  // returns; one that comes later creates a fresh lock file)
  ::unlink(lock.c_str());
  ::flock(lfd, LOCK_UN);
  ::close(lfd);
  if (result.empty() && why.empty())
    why = "the process that held " + lock + " did not produce the core (its compiler failed)";
  return result;
}

// what a core reports about itself: the hash of the sources it was compiled from, the size of its function table
struct CoreAbi {
  uint64_t src_hash;
  uint64_t api_bytes;
};

inline const nepmi_api* load_core(const std::string& path, uint64_t api_bytes, std::string& why)
{
  void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!h) {
    why = std::string("dlopen: ") + dlerror();
    return nullptr;
  }
  typedef const nepmi_api* (*get_api)(void);
  typedef CoreAbi (*get_abi)(void);
  get_api f = reinterpret_cast<get_api>(dlsym(h, "nepmi_core_api"));
  get_abi g = reinterpret_cast<get_abi>(dlsym(h, "nepmi_core_abi"));
  if (!f || !g) {
    why = "no nepmi_core_api / nepmi_core_abi in " + path;
    dlclose(h);
    return nullptr;
  }
  const CoreAbi abi = g();
  if (abi.src_hash != baked_hash() || abi.api_bytes != api_bytes) {
    why = path + " was built from other sources than this library (its name says otherwise: remove it)";
    dlclose(h);
    return nullptr;
  }
  return f();
}

// the core that serves models of this shape: loaded once per process; nullptr = the run-time-shape kernels of this library
inline const nepmi_api* core_for(const ShapeKey& key, uint64_t api_bytes)
{
  static std::mutex mu;
  static std::map<std::string, const nepmi_api*> cores;
  std::lock_guard<std::mutex> lk(mu);
  auto it = cores.find(key.name());
  if (it != cores.end())
    return it->second;
  const nepmi_api* api = nullptr;
  std::string why;
  const char* mode = std::getenv("NEPMI_JIT");
  const bool may_build = !(mode && mode[0] == '2'); // NEPMI_JIT=2: cores that exist already, never the compiler
  if (baked_hash() == 0) {
    why = "this library was built without a source hash (NEPMI_SRC_HASH: use the Makefile)";
  } else {
    const std::string name = core_name(key);
    std::string path;
    const std::string user = cache_dir();
    if (file_exists(lib_dir() + "/jit/" + name)) // cores built ahead of time, next to the library
      path = lib_dir() + "/jit/" + name;
    else if (!user.empty() && private_dir(user) && file_exists(user + "/" + name))
      path = user + "/" + name;
    if (path.empty() && may_build)
      path = build_core(key, user, why);
    if (!path.empty())
      api = load_core(path, api_bytes, why);
  }
  if (!api)
    std::fprintf(stderr, "nepmi: no JIT core for this model's shape (%s)\n", why.empty() ? "no core found, NEPMI_JIT=2" : why.c_str());
  if (api || may_build) // (a failed compilation is not tried again by this process; a look-up without the compiler may be)
    cores[key.name()] = api;
  return api;
}

} // namespace jit
} // namespace nepmi
