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

// ---- the output formats, restated from the reference's printf calls (src/main_cc/search_tool.cc; the tool itself needs
// flex/bison + RE2 and cannot be built in this image, so the formats are restated, not diffed against its output) ----------
struct Format { const char* name; const char* fmt; const char* ref; };
static const Format kFormats[] = {
    {"matches_row_head", "% 4" PRIi64 " \"", "search_tool.cc:1083"},      // then the matched string (fprint_alpha), then ...
    {"matches_row_tail", "\"%c", "search_tool.cc:1085"},                   // ... the closing quote and the separator
    {"total", "% 4" PRIi64 " total matches%c", "search_tool.cc:1112"},
    {"doc_info", "%.*s", "search_tool.cc:478"},                            // a matching document's info string
    {"doc_sep", "%c%s", "search_tool.cc:477"},                             // between documents: separator + prefix (prefix = "")
    {"offsets_lead", "%c\t", "search_tool.cc:480"},                        // --offsets: after the info string
    {"offset", " %" PRIi64, "search_tool.cc:499"},
    {"list_end", "%c", "search_tool.cc:519"},                              // after the last document of a non-empty list
};
static const char* fmt_of(const char* name) {
  for (const Format& f : kFormats)
    if (!strcmp(f.name, name)) return f.fmt;
  abort();
}
static void print_matches_row(FILE* out, int64_t n, const char* s, size_t len, char sep) {
  fprintf(out, fmt_of("matches_row_head"), n);
  fwrite(s, 1, len, out);            // fprint_alpha of a literal pattern: its bytes
  fprintf(out, fmt_of("matches_row_tail"), sep);
}
static void print_total(FILE* out, int64_t total, char sep) { fprintf(out, fmt_of("total"), total, sep); }
// documents in document order, each with its offsets ascending; the reference prints the separator BEFORE every document but
// the first (doc_sep) and femto_search's caller ends the list with one more separator
static void print_documents(FILE* out, const std::vector<std::pair<std::string, std::vector<int64_t>>>& docs, bool offsets, char sep) {
  bool first = true;
  for (const auto& d : docs) {
    if (!first) fprintf(out, fmt_of("doc_sep"), sep, "");
    first = false;
    fprintf(out, fmt_of("doc_info"), int(d.first.size()), d.first.data());
    if (offsets) {
      fprintf(out, fmt_of("offsets_lead"), sep);
      for (int64_t o : d.second) fprintf(out, fmt_of("offset"), o);
    }
  }
  if (!docs.empty()) fprintf(out, fmt_of("list_end"), sep);
}
// --formats: the table above, one "name<TAB>format<TAB>reference line" per row; --format-selftest: a fixed result rendered
// through the very functions the tool prints with (tests/test_host_logic.py compares both with the formats restated there)
static int print_formats() {
  for (const Format& f : kFormats) printf("%s\t%s\t%s\n", f.name, f.fmt, f.ref);
  return 0;
}
static int format_selftest(char sep) {
  print_matches_row(stdout, 7, "the", 3, sep);
  print_matches_row(stdout, 12345, "a \"b\"", 5, sep);
  print_total(stdout, 12352, sep);
  print_documents(stdout, {{"doc0.txt", {3, 17, 4242}}, {"dir/doc1", {0}}}, true, sep);
  print_documents(stdout, {{"doc0.txt", {3, 17, 4242}}, {"dir/doc1", {0}}}, false, sep);
  print_documents(stdout, {}, true, sep);
  print_total(stdout, 0, sep);
  return 0;
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
    else if (a == "--formats") return print_formats();
    else if (a == "--format-selftest") return format_selftest(sep);
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
      if (matches && n > 0) print_matches_row(out, n, pattern.data(), pattern.size(), sep);
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
      std::vector<std::pair<std::string, std::vector<int64_t>>> docs;
      int64_t prev_doc = -1;
      for (size_t i = 0; i < hits.size(); i++) {
        if (hits[i].first != prev_doc) {
          const char* info = nullptr;
          int64_t len = 0;
          if ((rc = femto_amd_document_info(ix, hits[i].first, &info, &len))) die("femto_amd_document_info", rc);
          docs.emplace_back(std::string(info, size_t(len)), std::vector<int64_t>());
          prev_doc = hits[i].first;
        }
        docs.back().second.push_back(hits[i].second);
      }
      print_documents(out, docs, offsets, sep);
    }
    femto_amd_close(ix);
  }
  if (count) print_total(out, total_matches, sep);
  if (out != stdout) fclose(out);
  return 0;
}
