// ckpt.h -- the record stream behind ddrl_*_save_run / ddrl_*_load_previous_run: what
// the reference's save_run / load_previous_run (DeepQNAg.hpp:504-537, CMAESAg.hpp:279-308; --continue,
// Simulator.hpp:76-84,196-205) keep across a pre-emption, for a whole population in one file.
//
// A file is the magic "DDRLRUN2" (format version 2: round 3; files of version 1 are refused by the magic) followed by records [u32 tag length][tag][u64 payload bytes][payload].
// ONE traversal function per object both writes and reads (Ckpt::writing picks the direction), so the two
// cannot drift apart; on reading, every record's tag and size are checked against what the receiving
// object expects, and a file from a different configuration is refused with DDRL_EIO before anything of
// the object has been overwritten by it (the `meta` record comes first).  Little-endian, host layout:
// a checkpoint is for resuming on the same kind of machine, not an exchange format.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <unistd.h>
#include <fcntl.h>
#include <atomic>
#include <string>
#include <vector>
#include "../../include/ddrl_pop.h"
#include "host_util.h"

struct Ckpt {
  FILE *f = nullptr;
  bool writing = false;
  std::string path;
  hipStream_t st = nullptr;
  std::vector<char> stage;
  std::string tmp;
  uint64_t file_bytes = 0;
  ~Ckpt() { if (f) { fclose(f); if (writing) remove(tmp.c_str()); } }     // an abandoned write leaves no partial file

  int open(const char *p, bool w) {
    if (!p) return ddrl_fail(DDRL_EINVAL, "NULL path");
    path = p; writing = w;
    // written next to the target and renamed over it once complete and on disk (finish): a pre-emption in the middle
    // of a multi-GB write -- the event run files exist for -- leaves the previous file intact
    // (the temporary's name is this process's and this writer's own: two handles or ranks saving to one target do not share it)
    static std::atomic<unsigned> serial{0};
    tmp = path + ".tmp." + std::to_string((long)getpid()) + "." + std::to_string(serial.fetch_add(1));
    f = fopen(w ? tmp.c_str() : p, w ? "wb" : "rb");
    if (!f) return ddrl_fail(DDRL_EIO, "cannot open %s", w ? tmp.c_str() : p);
    if (!w) {                                                            // bounds every length a record may claim
      if (fseek(f, 0, SEEK_END) == 0) { const long e = ftell(f); file_bytes = e > 0 ? (uint64_t)e : 0; }
      rewind(f);
    }
    char magic[8] = {'D', 'D', 'R', 'L', 'R', 'U', 'N', '2'};
    if (w) { if (fwrite(magic, 1, 8, f) != 8) return ddrl_fail(DDRL_EIO, "write to %s failed", p); return DDRL_OK; }
    char got[8];
    if (fread(got, 1, 8, f) != 8 || memcmp(got, magic, 8) != 0) return ddrl_fail(DDRL_EIO, "%s is not a ddrl_pop run file", p);
    return DDRL_OK;
  }
  // record header; reading: *nbytes receives the stored size (fixed == true: it must equal the given one)
  int head(const char *tag, uint64_t *nbytes, bool fixed) {
    const uint32_t tl = (uint32_t)strlen(tag);
    if (writing) {
      if (fwrite(&tl, 4, 1, f) != 1 || fwrite(tag, 1, tl, f) != tl || fwrite(nbytes, 8, 1, f) != 1)
        return ddrl_fail(DDRL_EIO, "write to %s failed", path.c_str());
      return DDRL_OK;
    }
    uint32_t gl = 0;
    char gt[64];
    uint64_t gn = 0;
    if (fread(&gl, 4, 1, f) != 1 || gl >= sizeof(gt) || fread(gt, 1, gl, f) != gl || fread(&gn, 8, 1, f) != 1)
      return ddrl_fail(DDRL_EIO, "%s: truncated before record '%s'", path.c_str(), tag);
    gt[gl] = 0;
    if (gl != tl || memcmp(gt, tag, tl) != 0)
      return ddrl_fail(DDRL_EIO, "%s does not match this object: record '%s' where '%s' was expected", path.c_str(), gt, tag);
    if (fixed && gn != *nbytes)
      return ddrl_fail(DDRL_EIO, "%s does not match this object: record '%s' holds %llu bytes, %llu expected", path.c_str(), tag,
                       (unsigned long long)gn, (unsigned long long)*nbytes);
    *nbytes = gn;
    return DDRL_OK;
  }
  int payload(void *p, size_t n) {
    if (n == 0) return DDRL_OK;
    if (writing ? fwrite(p, 1, n, f) != n : fread(p, 1, n, f) != n)
      return ddrl_fail(DDRL_EIO, "%s: short %s", path.c_str(), writing ? "write" : "read");
    return DDRL_OK;
  }
  int host(const char *tag, void *p, size_t n) {
    uint64_t nb = n;
    int rc = head(tag, &nb, true);
    return rc ? rc : payload(p, n);
  }
  template <class T> int val(const char *tag, T &x) { return host(tag, &x, sizeof(T)); }
  template <class T> int vec(const char *tag, std::vector<T> &v) {          // the length travels with the record
    uint64_t nb = v.size() * sizeof(T);
    int rc = head(tag, &nb, false);
    if (rc) return rc;
    if (!writing) {
      if (nb % sizeof(T) || nb > file_bytes) return ddrl_fail(DDRL_EIO, "%s: record '%s' has a broken length", path.c_str(), tag);
      v.resize(nb / sizeof(T));
    }
    return payload(v.data(), nb);
  }
  // device memory, through a bounded host staging buffer on the object's stream (a population's replay
  // rings are gigabytes: they move in 64-MiB pieces)
  int dev(const char *tag, void *d, size_t n) {
    uint64_t nb = n;
    int rc = head(tag, &nb, true);
    if (rc || n == 0) return rc;
    if (!d) return ddrl_fail(DDRL_ESTATE, "checkpoint: no device buffer for record '%s'", tag);
    const size_t piece = (size_t)64 << 20;
    stage.resize(n < piece ? n : piece);
    for (size_t off = 0; off < n; off += piece) {
      const size_t m = n - off < piece ? n - off : piece;
      char *dp = (char *)d + off;
      if (writing) {
        if (hipMemcpyAsync(stage.data(), dp, m, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
          return ddrl_fail(DDRL_EHIP, "checkpoint: device read of '%s' failed", tag);
        if ((rc = payload(stage.data(), m))) return rc;
      } else {
        if ((rc = payload(stage.data(), m))) return rc;
        if (hipMemcpyAsync(dp, stage.data(), m, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
          return ddrl_fail(DDRL_EHIP, "checkpoint: device write of '%s' failed", tag);
      }
    }
    return DDRL_OK;
  }
  // a value that must be the same in the file and in the receiving object (shape, algorithm, sizes)
  int same(const char *tag, const void *expect, size_t n) {
    if (writing) return host(tag, const_cast<void *>(expect), n);
    std::vector<char> got(n);
    int rc = host(tag, got.data(), n);
    if (rc) return rc;
    if (memcmp(got.data(), expect, n) != 0)
      return ddrl_fail(DDRL_EIO, "%s was written by a different configuration ('%s' differs)", path.c_str(), tag);
    return DDRL_OK;
  }
  int finish() {
    if (writing && (fflush(f) != 0 || ferror(f) || fsync(fileno(f)) != 0)) return ddrl_fail(DDRL_EIO, "write to %s failed", tmp.c_str());
    int rc = DDRL_OK;
    if (fclose(f) != 0 && writing) rc = ddrl_fail(DDRL_EIO, "close of %s failed", tmp.c_str());
    f = nullptr;
    if (writing && rc == DDRL_OK && rename(tmp.c_str(), path.c_str()) != 0) rc = ddrl_fail(DDRL_EIO, "cannot rename %s to %s", tmp.c_str(), path.c_str());
    if (writing && rc != DDRL_OK) remove(tmp.c_str());
    if (writing && rc == DDRL_OK) {          // the rename itself reaches the disk with the directory (best effort: not every file system lets a directory be opened)
      const size_t slash = path.find_last_of('/');
      const std::string dir = slash == std::string::npos ? std::string(".") : slash == 0 ? std::string("/") : path.substr(0, slash);
      const int dfd = ::open(dir.c_str(), O_RDONLY);
      if (dfd >= 0) { (void)fsync(dfd); ::close(dfd); }
    }
    return rc;
  }
};
