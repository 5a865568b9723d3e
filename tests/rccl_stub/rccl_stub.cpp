// TEST INFRASTRUCTURE: a stand-in for librccl with the five entry points libsmj's collective path binds (csrc/smj_comm.h), so that
// smj_comm_open's id-file rendezvous and the all-gather call sequence run at world size > 1 on a box without GPUs.  "Device" buffers
// are host memory; ranks meet through files in a directory named after the unique id.  Nothing here is RCCL's algorithm.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <string>
#include <vector>

struct Id { char internal[128]; };
struct Comm { std::string dir; int rank, world; long seq; };

static void nap() { struct timespec ts = {0, 2 * 1000 * 1000}; nanosleep(&ts, nullptr); }
static bool put(const std::string& path, const void* p, size_t n) {
  std::string tmp = path + ".tmp";
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) return false;
  const bool ok = fwrite(p, 1, n, f) == n;
  fclose(f);
  return ok && rename(tmp.c_str(), path.c_str()) == 0;
}
static bool get(const std::string& path, void* p, size_t n, double timeout_s) {
  for (double waited = 0; waited < timeout_s; waited += 0.002) {
    FILE* f = fopen(path.c_str(), "rb");
    if (f) {
      const size_t got = fread(p, 1, n, f);
      fclose(f);
      if (got == n) return true;
    }
    nap();
  }
  return false;
}

extern "C" {
int ncclGetUniqueId(Id* id) {
  memset(id, 0, sizeof *id);
  FILE* f = fopen("/dev/urandom", "rb");
  unsigned char r[16] = {0};
  if (f) { if (fread(r, 1, 16, f) != 16) r[0] = 1; fclose(f); }
  char* p = id->internal;
  p += sprintf(p, "stub-");
  for (int i = 0; i < 16; i++) p += sprintf(p, "%02x", r[i]);
  return 0;
}
int ncclCommInitRank(Comm** out, int world, Id id, int rank) {
  if (strncmp(id.internal, "stub-", 5) != 0 || memchr(id.internal, 0, sizeof id.internal) == nullptr) return 5;   // ncclInvalidArgument: not an id this library made
  const char* base = getenv("SMJ_STUB_DIR");
  Comm* c = new Comm{std::string(base ? base : "/tmp") + "/" + id.internal, rank, world, 0};
  mkdir(c->dir.c_str(), 0700);
  char me[64];
  snprintf(me, sizeof me, "/join.%d", rank);
  int one = 1, got;
  if (!put(c->dir + me, &one, sizeof one)) { delete c; return 2; }
  for (int r = 0; r < world; r++) {   // every rank must show up: ncclCommInitRank is collective
    snprintf(me, sizeof me, "/join.%d", r);
    if (!get(c->dir + me, &got, sizeof got, 60.0)) { delete c; return 2; }
  }
  *out = c;
  return 0;
}
int ncclAllGather(const void* send, void* recv, size_t count, int dtype, Comm* c, void* /*stream*/) {
  if (!c || dtype != 7) return 5;
  const size_t nb = count * 4;
  char name[96];
  snprintf(name, sizeof name, "/ag.%ld.%d", c->seq, c->rank);
  if (!put(c->dir + name, send, nb)) return 2;
  for (int r = 0; r < c->world; r++) {
    snprintf(name, sizeof name, "/ag.%ld.%d", c->seq, r);
    if (!get(c->dir + name, (char*)recv + (size_t)r * nb, nb, 60.0)) return 2;
  }
  c->seq++;
  return 0;
}
int ncclCommDestroy(Comm* c) { delete c; return 0; }
const char* ncclGetErrorString(int r) { return r == 0 ? "no error" : r == 5 ? "invalid argument (stub)" : "system error (stub)"; }
}
