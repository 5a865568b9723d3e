// TEST INFRASTRUCTURE: one rank of a world-size-N job driving the product's collective code (stretch_mujoco_amd/csrc/smj_comm.h: the
// file smj_capi.hip's smj_comm_init / smj_allgather_returns / smj_comm_destroy wrap with hipSetDevice) on host memory.
//   comm_harness <rank> <world> <id_path> <timeout_s> <count> [<rounds>]
// prints one line "OK <world*count values of the last gather>" or "ERR <rc> <message>".
#include "../../stretch_mujoco_amd/csrc/smj_comm.h"

#include <vector>

int main(int argc, char** argv) {
  if (argc < 6) return 2;
  const int rank = atoi(argv[1]), world = atoi(argv[2]), count = atoi(argv[5]), rounds = argc > 6 ? atoi(argv[6]) : 1;
  const char* path = argv[3];
  const double timeout = atof(argv[4]);
  SmjComm cm;
  std::string err;
  int rc = smj_comm_open(cm, rank, world, *path ? path : nullptr, timeout, err);
  if (rc) { printf("ERR %d %s\n", rc, err.c_str()); return 1; }
  std::vector<float> send(count), recv((size_t)world * count, -1.f);
  for (int k = 0; k < rounds; k++) {
    for (int i = 0; i < count; i++) send[i] = 1000.f * rank + i + 0.25f * k;   // per-env returns of this rank's shard
    rc = smj_comm_allgather(cm, send.data(), recv.data(), count, nullptr, err);
    if (rc) { printf("ERR %d %s\n", rc, err.c_str()); return 1; }
  }
  printf("OK");
  for (float v : recv) printf(" %g", v);
  printf("\n");
  rc = smj_comm_open(cm, rank, world, path, timeout, err);   // a second init on a live communicator is refused
  if (rc != -1) { printf("ERR second init returned %d\n", rc); return 1; }
  smj_comm_close(cm);
  return 0;
}
