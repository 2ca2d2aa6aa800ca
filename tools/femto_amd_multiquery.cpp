// femto_amd_multiquery -- C++ host program over the C ABI (include/femto_amd.h), the counterpart of the
// reference's batch harness femto_multiquery (src/main/query_tool.c): same command line shape
//     femto_amd_multiquery <index path> -count            < queries
//     femto_amd_multiquery <index path> -locate [max]     < queries
// same Pizza&Chili query format on stdin (one line "# number=<n> length=<m> ..." followed by n*m raw
// pattern bytes, query_tool.c:48-98) and the same kind of report ("Did N parallel count queries in: ...
// That's X .../sec", print_timings of src/utils/timing.c:104-134, plus "Counted N results").
// Extras: --device N selects the GPU, --dump FILE writes the raw results (count: i64 first[n], i64 last[n];
// locate: i32 noccs[n] then all i64 offsets) for scripted comparison.
#include <chrono>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/femto_amd.h"

static void die(const char* what, int rc) {
  fprintf(stderr, "%s failed: error %d: %s\n", what, rc, femto_amd_last_error());
  exit(1);
}

static void report(const char* thing, double number, double seconds) {
  printf("Did %f %s in:\n", number, thing);
  printf("   Real time: %f\n", seconds);
  printf(" That's %g %s/sec or %g sec average\n", number / seconds, thing, seconds / number);
}

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s <index path> -count | -locate [max] [--device N] [--dump FILE] < queries\n", argv[0]);
    return 2;
  }
  const char* index_path = argv[1];
  const std::string todo = argv[2];
  int max_occs = INT32_MAX;
  int device = 0;
  const char* dump = nullptr;
  for (int i = 3; i < argc; i++) {
    if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--dump") && i + 1 < argc) dump = argv[++i];
    else max_occs = atoi(argv[i]);
  }

  printf("Reading queries\n");
  char hdr[1024];
  if (!fgets(hdr, sizeof hdr, stdin) || hdr[0] != '#') {
    fprintf(stderr, "Bad query file format; missing #\n");
    return 1;
  }
  int number = 0, length = 0;
  if (sscanf(hdr, "# number=%i length=%i ", &number, &length) != 2 || number < 0 || length < 0) {
    fprintf(stderr, "Bad query file format in header line\n");
    return 1;
  }
  std::vector<uint8_t> raw(size_t(number) * size_t(length));
  if (!raw.empty() && fread(raw.data(), 1, raw.size(), stdin) != raw.size()) {
    fprintf(stderr, "Bad query file format: short pattern data\n");
    return 1;
  }
  std::vector<int32_t> plen(size_t(number), length);
  std::vector<int64_t> starts((size_t(number)));
  for (int i = 0; i < number; i++) starts[size_t(i)] = int64_t(i) * length;
  std::vector<uint16_t> codes(raw.size() + 1);
  for (size_t i = 0; i < raw.size(); i++) codes[i] = uint16_t(raw[i]) + FEMTO_AMD_CHARACTER_OFFSET;

  femto_amd_index_t* ix = nullptr;
  int rc = femto_amd_open(index_path, device, &ix);
  if (rc) die("femto_amd_open", rc);

  FILE* df = dump ? fopen(dump, "wb") : nullptr;
  if (todo == "-count") {
    printf("Starting count\n");
    std::vector<int64_t> first((size_t(number))), last((size_t(number)));
    const auto t0 = std::chrono::steady_clock::now();
    rc = femto_amd_count_flat(ix, number, plen.data(), codes.data(), starts.data(), first.data(), last.data());
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rc) die("femto_amd_count_flat", rc);
    int64_t nresults = 0;
    for (int i = 0; i < number; i++) nresults += last[size_t(i)] - first[size_t(i)] + 1;
    report("parallel count queries", number, dt);
    printf("Counted %" PRIi64 " results\n", nresults);
    if (df) {
      fwrite(first.data(), 8, size_t(number), df);
      fwrite(last.data(), 8, size_t(number), df);
    }
  } else if (todo == "-locate") {
    std::vector<int32_t> noccs((size_t(number)));
    std::vector<int64_t> ostarts(size_t(number) + 1);
    int64_t total = 0;
    const auto t0 = std::chrono::steady_clock::now();
    rc = femto_amd_locate_flat(ix, number, plen.data(), codes.data(), starts.data(), max_occs, noccs.data(), ostarts.data(),
                               nullptr, 0, &total);
    if (rc) die("femto_amd_locate_flat", rc);
    std::vector<int64_t> offsets(size_t(total) + 1);
    if (total) {
      rc = femto_amd_locate_flat(ix, number, plen.data(), codes.data(), starts.data(), max_occs, noccs.data(), ostarts.data(),
                                 offsets.data(), total, &total);
      if (rc) die("femto_amd_locate_flat", rc);
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    report("parallel locate queries", number, dt);
    report("parallel locate results", double(total), dt);
    if (df) {
      fwrite(noccs.data(), 4, size_t(number), df);
      fwrite(offsets.data(), 8, size_t(total), df);
    }
  } else {
    fprintf(stderr, "unknown mode %s (use -count or -locate [max])\n", todo.c_str());
    return 2;
  }
  if (df) fclose(df);
  femto_amd_close(ix);
  return 0;
}
