// rccl_selftest.cpp -- the sharded-BA exchange through RCCL (include/sfmhip_rccl.h) in a C++ program:
// what the reference's single-process orchestration would become with one process per GPU.
//
//   rccl_selftest <problem.bin> <out.bin> [id-file rank world]
// problem.bin: n_cam n_pt n_obs iters (int32), cams6, pts3, focal, obs_cam, obs_pt, obs_xy (all points: every rank
// reads the same file and keeps the points p with p % world == rank, with their observations).
// One rank: a 1-rank communicator declared as rank 0 of a 2-rank job whose other rank holds no points --
// the pack / ncclAllReduce / unpack path runs and must reproduce the plain single-process solve.
// out.bin (rank 0): cams6, focal, final cost, iterations.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <vector>
#include "../../../include/sfmhip_rccl.h"

template <typename T>
static std::vector<T> rdv(FILE* f, size_t n) {
  std::vector<T> v(n);
  if (n && fread(v.data(), sizeof(T), n, f) != n) {
    fprintf(stderr, "short read\n");
    exit(2);
  }
  return v;
}
#define CK(x)                                                                         \
  do {                                                                                \
    const int rc_ = (x);                                                              \
    if (rc_ != SFMHIP_OK) {                                                           \
      fprintf(stderr, "%s -> %d (%s) nccl %d\n", #x, rc_, sfmhip_error_string(rc_), sfmhip_rccl_last_error()); \
      return 1;                                                                       \
    }                                                                                 \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const char* id_file = argc > 5 ? argv[3] : nullptr;
  const int rank = argc > 5 ? atoi(argv[4]) : 0, world = argc > 5 ? atoi(argv[5]) : 1;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  const std::vector<int> hd = rdv<int>(f, 4);
  const int n_cam = hd[0], n_pt = hd[1], n_obs = hd[2], iters = hd[3];
  std::vector<double> cams = rdv<double>(f, 6 * (size_t)n_cam), pts = rdv<double>(f, 3 * (size_t)n_pt);
  const double focal = rdv<double>(f, 1)[0];
  const std::vector<int> ocam = rdv<int>(f, n_obs), opt = rdv<int>(f, n_obs);
  const std::vector<double> oxy = rdv<double>(f, 2 * (size_t)n_obs);
  fclose(f);
  // this rank's shard: points p % world == rank, renumbered
  std::vector<int> local(n_pt, -1);
  std::vector<double> lpts;
  for (int p = 0; p < n_pt; ++p)
    if (p % world == rank) {
      local[p] = (int)(lpts.size() / 3);
      lpts.insert(lpts.end(), pts.begin() + 3 * p, pts.begin() + 3 * p + 3);
    }
  std::vector<int> lcam, lpt;
  std::vector<double> lxy;
  for (int o = 0; o < n_obs; ++o)
    if (local[opt[o]] >= 0) {
      lcam.push_back(ocam[o]);
      lpt.push_back(local[opt[o]]);
      lxy.push_back(oxy[2 * o]);
      lxy.push_back(oxy[2 * o + 1]);
    }

  sfmhip_ctx* ctx = nullptr;
  const int n_dev = sfmhip_device_count();  // one rank per GPU (RCCL refuses two ranks on one device)
  if (n_dev <= 0) return 3;
  CK(sfmhip_init(rank % n_dev, &ctx));
  unsigned char id[SFMHIP_RCCL_ID_BYTES];
  if (rank == 0) {
    CK(sfmhip_rccl_unique_id(id));
    if (id_file) {
      FILE* g = fopen(id_file, "wb");
      fwrite(id, 1, sizeof id, g);
      fclose(g);
    }
  } else {
    for (int t = 0; t < 600; ++t) {  // wait for rank 0's id
      FILE* g = fopen(id_file, "rb");
      if (g) {
        const size_t n = fread(id, 1, sizeof id, g);
        fclose(g);
        if (n == sizeof id) break;
      }
      usleep(100000);
    }
  }
  void* comm = nullptr;
  CK(sfmhip_rccl_comm_create(ctx, rank, world, id, &comm));
  sfmhip_ba* ba = nullptr;
  CK(sfmhip_ba_create(ctx, n_cam, (int)(lpts.size() / 3), (int)lcam.size(), lcam.data(), lpt.data(), lxy.data(), &ba));
  // (one rank: declared as rank 0 of 2 so that the exchange runs; the communicator's sum is then this rank's part)
  CK(sfmhip_ba_use_rccl(ba, ctx, comm, rank, world == 1 ? 2 : world));
  CK(sfmhip_ba_set_params(ba, cams.data(), lpts.data(), focal));
  sfmhip_ba_summary sm;
  CK(sfmhip_ba_iterate(ba, iters, &sm));
  double f_out = 0;
  CK(sfmhip_ba_get_params(ba, cams.data(), lpts.data(), &f_out));
  if (rank == 0) {
    FILE* o = fopen(argv[2], "wb");
    fwrite(cams.data(), 8, cams.size(), o);
    fwrite(&f_out, 8, 1, o);
    fwrite(&sm.final_cost, 8, 1, o);
    fwrite(&sm.iterations, 4, 1, o);
    fclose(o);
  }
  sfmhip_ba_destroy(ba);
  sfmhip_rccl_comm_destroy(comm);
  sfmhip_shutdown(ctx);
  return 0;
}
