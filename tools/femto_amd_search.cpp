// femto_amd_search -- command-line search over the C ABI (include/femto_amd.h), shaped like the reference's
// femto_search (src/main_cc/search_tool.cc) for LITERAL patterns so that scripts can swap binaries:
//     femto_amd_search [options] <index_path> [<index_path>...] <pattern>
//       --count             print only "% 4d total matches"                 (search_tool.cc:1108-1113)
//       --matches           with the count, one row '% 4d "pattern"' per index that matches (:1083-1087)
//       --offsets           list matching documents with the offsets of the matches      (:470-502, :517-522)
//       (neither)           list matching documents
//       --max_results <n>   at most n located matches per index (default 1048576, :544)
//       --null              separate output records with 0 bytes instead of newlines
//       --output <file>     write results to file instead of stdout
//       --pattern <p> | --pattern-from <file>
//       --device <n>        GPU to use (extension)
// The reference parses <pattern> as a regular expression and runs it on its CPU scheduler; this tool takes the
// pattern literally and runs femto_amd_count_flat / femto_amd_locate_flat on the GPU.  A pattern that contains
// a regular-expression metacharacter is refused unless --literal is given, so that the two tools never
// silently disagree.  Documents are listed in document order, offsets ascending (the order of the reference's
// result sets, src/main/results.h) as "<info><sep>\t <off> <off>...<sep>".
#include <algorithm>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../include/femto_amd.h"

static void die(const char* what, int rc) {
  fprintf(stderr, "%s failed: error %d: %s\n", what, rc, femto_amd_last_error());
  exit(1);
}

static void usage(const char* name) {
  printf("Usage: %s [options] <index_path> [<index_path>...] <pattern>\n", name);
  printf(" where options include:\n");
  printf(" --max_results <number> set the maximum number of results\n");
  printf(" --offsets request document offsets\n");
  printf(" --count Ask for only the number of results\n");
  printf(" --matches Show the matching string with its count\n");
  printf(" --output <filename> output query results to file instead of stdout\n");
  printf(" --null seperate output lines with 0 bytes instead of newlines\n");
  printf(" --pattern <argument> pattern in argument (by default the pattern is the last non-option argument)\n");
  printf(" --pattern-from <filename> read pattern from filename instead of intepreting it as the pattern\n");
  printf(" --literal take regular-expression metacharacters in the pattern literally\n");
  printf(" --device <number> GPU to run on\n");
  exit(2);
}

int main(int argc, char** argv) {
  std::vector<std::string> paths;
  std::string pattern;
  bool have_pattern = false, offsets = false, count = false, matches = false, literal = false;
  int64_t max_results = 1024 * 1024;
  const char* output = nullptr;
  char sep = '\n';
  int device = 0;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    auto next = [&]() -> const char* {
      if (i + 1 >= argc) usage(argv[0]);
      return argv[++i];
    };
    if (a == "--max_results") max_results = strtoll(next(), nullptr, 0);
    else if (a == "--offsets") offsets = true;
    else if (a == "--count") count = true;
    else if (a == "--matches") matches = count = true;
    else if (a == "--output") output = next();
    else if (a == "--null") sep = '\0';
    else if (a == "--literal") literal = true;
    else if (a == "--device") device = atoi(next());
    else if (a == "--pattern") { pattern = next(); have_pattern = true; }
    else if (a == "--pattern-from") {
      const char* fn = next();
      FILE* f = fopen(fn, "rb");
      if (!f) { printf("Could not read pattern from %s\n", fn); return 1; }
      char buf[4096];
      size_t k;
      while ((k = fread(buf, 1, sizeof buf, f)) > 0) pattern.append(buf, k);
      fclose(f);
      have_pattern = true;
    } else if (a.size() > 1 && a[0] == '-' && a[1] == '-') {
      printf("Unknown option %s\n", a.c_str());
      usage(argv[0]);
    } else paths.push_back(a);
  }
  if (!have_pattern) {
    if (paths.empty()) usage(argv[0]);
    pattern = paths.back();
    paths.pop_back();
  }
  if (paths.empty()) usage(argv[0]);
  if (max_results < 0 || max_results > INT32_MAX) max_results = INT32_MAX;
  if (!literal && pattern.find_first_of(".*+?|()[]{}\\^$") != std::string::npos) {
    fprintf(stderr, "Could not parse pattern %s: regular expressions are not supported by this tool "
                    "(use --literal to search for the characters themselves)\n", pattern.c_str());
    return 1;
  }
  FILE* out = stdout;
  if (output && !(out = fopen(output, "w"))) {
    fprintf(stderr, "Could not open output filename '%s' for writing\n", output);
    return 1;
  }

  std::vector<uint16_t> pat(pattern.size() ? pattern.size() : 1);
  for (size_t i = 0; i < pattern.size(); i++) pat[i] = uint16_t(uint8_t(pattern[i])) + FEMTO_AMD_CHARACTER_OFFSET;
  const int32_t plen = int32_t(pattern.size());
  const int64_t start = 0;

  int64_t total_matches = 0;
  for (const std::string& path : paths) {
    femto_amd_index_t* ix = nullptr;
    int rc = femto_amd_open(path.c_str(), device, &ix);
    if (rc) {
      printf("Could not open index at %s\n", path.c_str());
      die("femto_amd_open", rc);
    }
    if (count) {
      int64_t first = 0, last = -1;
      if ((rc = femto_amd_count_flat(ix, 1, &plen, pat.data(), &start, &first, &last))) die("femto_amd_count_flat", rc);
      const int64_t n = last >= first ? last - first + 1 : 0;
      if (matches && n > 0) {
        fprintf(out, "% 4" PRIi64 " \"", n);
        fwrite(pattern.data(), 1, pattern.size(), out);
        fprintf(out, "\"%c", sep);
      }
      total_matches += n;
    } else {
      int32_t noccs = 0;
      int64_t ostarts[2] = {0, 0}, total = 0;
      if ((rc = femto_amd_locate_flat(ix, 1, &plen, pat.data(), &start, int(max_results), &noccs, ostarts, nullptr, 0, &total)))
        die("femto_amd_locate_flat", rc);
      std::vector<int64_t> offs(size_t(total ? total : 1));
      if (total && (rc = femto_amd_locate_flat(ix, 1, &plen, pat.data(), &start, int(max_results), &noccs, ostarts, offs.data(),
                                               total, &total)))
        die("femto_amd_locate_flat", rc);
      std::vector<std::pair<int64_t, int64_t>> hits;  // (document, offset in document)
      for (int64_t i = 0; i < total; i++) {
        int64_t doc = 0, doff = 0;
        if ((rc = femto_amd_resolve_location(ix, offs[size_t(i)], &doc, &doff))) die("femto_amd_resolve_location", rc);
        hits.emplace_back(doc, doff);
      }
      std::sort(hits.begin(), hits.end());
      int64_t prev_doc = -1;
      for (size_t i = 0; i < hits.size(); i++) {
        if (hits[i].first != prev_doc) {
          if (prev_doc != -1) fputc(sep, out);
          const char* info = nullptr;
          int64_t len = 0;
          if ((rc = femto_amd_document_info(ix, hits[i].first, &info, &len))) die("femto_amd_document_info", rc);
          fwrite(info, 1, size_t(len), out);
          if (offsets) fprintf(out, "%c\t", sep);
          prev_doc = hits[i].first;
        }
        if (offsets) fprintf(out, " %" PRIi64, hits[i].second);
      }
      if (!hits.empty()) fputc(sep, out);
    }
    femto_amd_close(ix);
  }
  if (count) fprintf(out, "% 4" PRIi64 " total matches%c", total_matches, sep);
  if (out != stdout) fclose(out);
  return 0;
}
