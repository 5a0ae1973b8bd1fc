// Host-only check program for f5-tts_amd/csrc/bigvgan_host.h (g++, no HIP): raw float32 in on stdin, raw float32 out on stdout.
//   filter                         -> 12 floats
//   taps  k u                      -> text "shift0 ntaps"
//   conv  cout cin k cpad          -> [cout, k*cpad]
//   convt cin cout k u cpad        -> [u*cout, ntaps*cpad]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "bigvgan_host.h"

static std::vector<float> read_floats(size_t n) {
  std::vector<float> v(n);
  if (fread(v.data(), sizeof(float), n, stdin) != n) { fprintf(stderr, "short read\n"); exit(2); }
  return v;
}

int main(int argc, char** argv) {
  if (argc < 2) return 1;
  if (!strcmp(argv[1], "filter")) {
    float f[12];
    bv_kaiser_sinc_12(f);
    fwrite(f, sizeof(float), 12, stdout);
  } else if (!strcmp(argv[1], "taps") && argc == 4) {
    int s0, nt;
    bv_convt_taps(atoi(argv[2]), atoi(argv[3]), s0, nt);
    printf("%d %d\n", s0, nt);
  } else if (!strcmp(argv[1], "conv") && argc == 6) {
    const int cout = atoi(argv[2]), cin = atoi(argv[3]), k = atoi(argv[4]), cpad = atoi(argv[5]);
    std::vector<float> w = read_floats((size_t)cout * cin * k), m;
    bv_conv_matrix(w.data(), cout, cin, k, cpad, m);
    fwrite(m.data(), sizeof(float), m.size(), stdout);
  } else if (!strcmp(argv[1], "convt") && argc == 7) {
    const int cin = atoi(argv[2]), cout = atoi(argv[3]), k = atoi(argv[4]), u = atoi(argv[5]), cpad = atoi(argv[6]);
    std::vector<float> w = read_floats((size_t)cin * cout * k), m;
    int s0, nt;
    bv_convt_taps(k, u, s0, nt);
    bv_convt_matrix(w.data(), cin, cout, k, u, cpad, s0, nt, m);
    fwrite(m.data(), sizeof(float), m.size(), stdout);
  } else {
    return 1;
  }
  return 0;
}
