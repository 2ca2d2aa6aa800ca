// femto_amd_search -- femto_search's counterpart over the C ABI (include/femto_amd.h): the same command line, the same
// query language, the same output, answered by the GPU, so that scripts (and the reference's end-to-end test,
// src/test/test.pl) can swap binaries:
//     femto_amd_search [options] <index_path> [<index_path>...] <pattern>
// What it follows, line by line, is src/main_cc/search_tool.cc:
//   * options (:566-633): -v/--verbose, --by_index, --max_results, --offsets, --count, --matches, --output, --null, --json,
//     --icase, --pattern, --pattern-from, --raw-pattern, --raw-pattern-from; the pattern is the last non-option argument
//     unless one of the pattern options gave it;
//   * the pattern is a QUERY in femto's language (src/main/QUERY_FORMAT.txt; femto_amd_query_compile restates the flex/bison
//     front end, streamline_query and simplify_query): a query that comes down to one string is searched as a string
//     (femto_amd_count_flat), everything else as an automaton (femto_amd_nfa_search_batch = do_regexp_query);
//   * --count / --matches (:905-1100): one row '% 4d "matched string"' per DISTINCT matched string -- longest first, then by
//     alpha code (matchcmp, :118-127), the same string found in several indexes summed -- and "% 4d total matches";
//   * otherwise (:1101-1110, print_matches :352-520): the matching documents in document order, with --offsets each followed
//     by "<sep>\t" and its offsets ascending.  Which rows are turned into documents is the reference's first result chunk
//     (do_string_results_query / do_regexp_results_query / do_range_to_results_query, src/main/server.c:4549-5220): the
//     result ranges in sorted order, each read in pieces of at most --max_results rows, until max_results or more rows
//     have been collected;
//   * --json (:889-894, :1075-1082, :1105-1114) wraps either form.
// Not provided, and refused by name rather than misread: --grep / --multigrep / --grepdir (they read the indexed files),
// --suggest / --suggest-starts, --filter-results (RE2), and the boolean operators of the query language (document-level
// result sets; SURVEY.md 8 "out of scope").  Extensions: --device <n>, --literal (take the pattern's bytes as they are: what
// --raw-pattern does, for scripts written against round 4's tool), --formats / --format-selftest (test hooks).
// Documents-only mode on an index WITH document chunks: the reference reads whole chunks' document lists
// (BLOCK_CHUNK_REQUEST_DOCUMENTS) and counts documents, not rows, against --max_results; this tool locates rows in both
// modes, so the two only differ when more than max_results (default 2^20) rows match.
#include <algorithm>
#include <cctype>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/stat.h>
#include <string>
#include <utility>
#include <vector>

#include "../include/femto_amd.h"

static void die(const char* what, int rc) {
  fprintf(stderr, "%s failed: error %d: %s\n", what, rc, femto_amd_last_error());
  exit(255);
}

// ---- the output formats, restated from the reference's printf calls (src/main_cc/search_tool.cc) --------------------------
struct Format { const char* name; const char* fmt; const char* ref; };
static const Format kFormats[] = {
    {"matches_row_head", "% 4" PRIi64 " \"", "search_tool.cc:1084"},      // then the matched string (fprint_alpha), then ...
    {"matches_row_tail", "\"%c", "search_tool.cc:1086"},                   // ... the closing quote and the separator
    {"total", "% 4" PRIi64 " total matches%c", "search_tool.cc:1112"},
    {"doc_info", "%.*s", "search_tool.cc:478"},                            // a matching document's info string
    {"doc_sep", "%c%s", "search_tool.cc:477"},                             // between documents: separator + prefix (prefix = "")
    {"offsets_lead", "%c\t", "search_tool.cc:480"},                        // --offsets: after the info string
    {"offset", " %" PRIi64, "search_tool.cc:499"},
    {"list_end", "%c", "search_tool.cc:519"},                              // after the last document of a non-empty list
    {"by_index_head", "Results from %s\n", "search_tool.cc:1038"},
    {"by_index_row", "% 4" PRIi64 " [%" PRIi64 ",%" PRIi64 "] \"", "search_tool.cc:1046"},
    {"json_open", "{\n \"pattern\":\"", "search_tool.cc:890-891"},
    {"json_results", "\",\n \"results\":[\n   ", "search_tool.cc:893"},
    {"json_row_sep", ",\n   ", "search_tool.cc:1077"},
    {"json_total", ",\n \"total\":%" PRIi64, "search_tool.cc:1111"},
    {"json_close", "\n}\n", "search_tool.cc:1114"},
};
static const char* fmt_of(const char* name) {
  for (const Format& f : kFormats)
    if (!strcmp(f.name, name)) return f.fmt;
  abort();
}
// alphatos (src/main/index_types.h:104-118): how one alpha code is printed
static std::string alpha_to_s(int alpha) {
  char buf[16];
  const int ch = alpha - FEMTO_AMD_CHARACTER_OFFSET;
  if (ch < 0) snprintf(buf, sizeof buf, "\\x-%02x", -ch);
  else if (ch == '\\' || ch == '"') snprintf(buf, sizeof buf, "\\%c", ch);
  else if (ch >= 32 && ch < 127) snprintf(buf, sizeof buf, "%c", ch);
  else snprintf(buf, sizeof buf, "\\x%02x", ch);
  return buf;
}
static void fprint_alpha(FILE* out, const std::vector<uint16_t>& s) {
  for (uint16_t a : s) fputs(alpha_to_s(a).c_str(), out);
}
// encode_ch_json (src/main/json.c:35-62)
static void fprint_ch_json(FILE* out, int ch) {
  if (ch == '"') fputs("\\\"", out);
  else if (ch == '\\') fputs("\\\\", out);
  else if (ch >= 32 && ch < 127) fputc(ch, out);
  else fprintf(out, "\\u%04x", ch);
}
static void fprint_alpha_json(FILE* out, const std::vector<uint16_t>& s) {      // encode_alpha_json: alphatos, then JSON
  for (uint16_t a : s)
    for (char c : alpha_to_s(a)) fprint_ch_json(out, (unsigned char)c);
}
static void print_matches_row(FILE* out, int64_t n, const std::vector<uint16_t>& s, char sep) {
  fprintf(out, fmt_of("matches_row_head"), n);
  fprint_alpha(out, s);
  fprintf(out, fmt_of("matches_row_tail"), sep);
}
static void print_total(FILE* out, int64_t total, char sep) { fprintf(out, fmt_of("total"), total, sep); }

using DocList = std::vector<std::pair<std::string, std::vector<int64_t>>>;
// print_matches (search_tool.cc:352-520) for one index's result set: the separator BEFORE every document but the first,
// one more after a non-empty list; JSON: [ ["info"], [offsets] ] rows ('|' in an info string splits it, GLOM_CHAR)
static void print_documents(FILE* out, const DocList& docs, bool offsets, char sep, bool json, bool* first_match) {
  bool first_keep = true;
  for (const auto& d : docs) {
    if (json) {
      if (!first_keep) fputs("] ],\n   ", out);
      else if (!*first_match) fputs(fmt_of("json_row_sep"), out);
      *first_match = false;
      fputs("[ [\"", out);
      for (char c : d.first) {
        if (c == '|') fputs("\",\"", out);
        else fprint_ch_json(out, (unsigned char)c);
      }
      fputs("\"], [", out);
      if (offsets)
        for (size_t k = 0; k < d.second.size(); k++) fprintf(out, k ? ", %" PRIi64 : "%" PRIi64, d.second[k]);
    } else {
      if (!first_keep) fprintf(out, fmt_of("doc_sep"), sep, "");
      fprintf(out, fmt_of("doc_info"), int(d.first.size()), d.first.data());
      if (offsets) {
        fprintf(out, fmt_of("offsets_lead"), sep);
        for (int64_t o : d.second) fprintf(out, fmt_of("offset"), o);
      }
    }
    first_keep = false;
  }
  if (!docs.empty()) {
    if (json) fputs(" ] ] ", out);
    else fprintf(out, fmt_of("list_end"), sep);
  }
}
static int print_formats() {
  for (const Format& f : kFormats) {
    std::string esc;
    for (const char* c = f.fmt; *c; c++) esc += *c == '\n' ? std::string("\\n") : *c == '\t' ? std::string("\\t") : std::string(1, *c);
    printf("%s\t%s\t%s\n", f.name, esc.c_str(), f.ref);
  }
  return 0;
}
static std::vector<uint16_t> alpha_of(const char* s) {
  std::vector<uint16_t> v;
  for (; *s; s++) v.push_back(uint16_t((unsigned char)*s) + FEMTO_AMD_CHARACTER_OFFSET);
  return v;
}
static int format_selftest(char sep) {
  bool first = true;
  print_matches_row(stdout, 7, alpha_of("the"), sep);
  print_matches_row(stdout, 12345, alpha_of("a \"b\"\\\x01\xff"), sep);
  print_matches_row(stdout, 1, std::vector<uint16_t>{2, 70}, sep);
  print_total(stdout, 12352, sep);
  print_documents(stdout, {{"doc0.txt", {3, 17, 4242}}, {"dir/doc1", {0}}}, true, sep, false, &first);
  print_documents(stdout, {{"doc0.txt", {3, 17, 4242}}, {"dir/doc1", {0}}}, false, sep, false, &first);
  print_documents(stdout, {}, true, sep, false, &first);
  print_total(stdout, 0, sep);
  first = true;
  print_documents(stdout, {{"a|b", {3, 17}}, {"c\"d", {0}}}, true, sep, true, &first);
  print_documents(stdout, {{"e", {5}}}, false, sep, true, &first);
  fputc('\n', stdout);
  fprint_alpha_json(stdout, alpha_of("q\"\\\x01"));
  fputc('\n', stdout);
  return 0;
}

static void usage(const char* name) {       // search_tool.cc:48-71, minus what is refused, plus --device / --literal
  printf("Usage: %s [options] <index_path> [<index_path>...] <pattern>\n", name);
  printf(" where options include:\n");
  printf(" -v or --verbose  print extra verbose output\n");
  printf(" --max_results <number> set the maximum number of results\n");
  printf(" --offsets request document offsets\n");
  printf(" --count Ask for only the number of results\n");
  printf(" --matches Show strings matching approximate search and/or regular expression\n");
  printf(" --output <filename> output query results to file instead of stdout\n");
  printf(" --null seperate output lines with 0 bytes instead of newlines\n");
  printf(" --icase make the search case-insensitive\n");
  printf(" --json output json\n");
  printf(" --pattern <argument> pattern in argument (by default the pattern is the last non-option argument\n");
  printf(" --pattern-from <filename> read pattern from filename instead of intepreting it as the pattern\n");
  printf(" --raw-pattern <argument> / --raw-pattern-from <filename> search for the bytes themselves\n");
  printf(" --by_index print results per searched index\n");
  printf(" --device <number> GPU to run on (femto_amd_search only)\n");
  printf(" --literal same as giving the pattern with --raw-pattern (femto_amd_search only)\n");
  printf("The pattern is a query in femto's language (regular expressions over bytes, APPROX; QUERY_FORMAT.txt),\n");
  printf("restated by hand from the reference's flex/bison grammar. Not supported: --grep --multigrep --grepdir\n");
  printf("--suggest --suggest-starts --filter-results, and the boolean operators AND OR NOT THEN WITHIN.\n");
  exit(255);
}
static void refuse(const char* opt) {
  fprintf(stderr, "Option %s is not supported by femto_amd_search\n", opt);
  exit(255);
}
static bool read_file(const char* fn, std::string* out) {
  FILE* f = fopen(fn, "rb");
  if (!f) return false;
  char buf[65536];
  size_t k;
  while ((k = fread(buf, 1, sizeof buf, f)) > 0) out->append(buf, k);
  fclose(f);
  return true;
}

struct Match {            // struct match_info (search_tool.cc:76-85)
  int64_t first, last;
  int cost;
  std::vector<uint16_t> match;
  int index;
};
// matchcmp (search_tool.cc:118-127) without suggestions: longer strings first, then alphacmp
static bool match_less(const Match& a, const Match& b) {
  if (a.match.size() != b.match.size()) return a.match.size() > b.match.size();
  return a.match < b.match;
}

// the string of `len` symbols every row of [first, ...] starts with: len LF^-1 steps from `first`, all results at once
static void matched_strings(femto_amd_index_t* ix, const std::vector<int64_t>& firsts, const std::vector<int32_t>& lens,
                            std::vector<std::vector<uint16_t>>* out) {
  const size_t n = firsts.size();
  out->assign(n, {});
  std::vector<int64_t> rows(firsts), next(n), off(n);
  std::vector<uint16_t> ch(n);
  int32_t longest = 0;
  for (int32_t l : lens) longest = std::max(longest, l);
  for (int32_t k = 0; k < longest; k++) {
    const int rc = femto_amd_forward_steps(ix, int64_t(n), rows.data(), ch.data(), next.data(), off.data());
    if (rc) die("femto_amd_forward_steps", rc);
    for (size_t i = 0; i < n; i++) {
      if (k < lens[i]) (*out)[i].push_back(ch[i]);
      if (next[i] >= 0) rows[i] = next[i];      // (a string that runs into an end-of-document marker stops stepping)
    }
  }
}

int main(int argc, char** argv) {
  std::vector<std::string> paths;
  std::string pattern, rawpattern;
  bool have_pattern = false, have_raw = false, offsets = false, count = false, matches = false, literal = false, icase = false;
  bool json = false, by_index = false;
  int verbose = 0;
  int64_t chunk_size = 1024 * 1024;
  const char* output = nullptr;
  char sep = '\n';
  int device = 0;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    auto next = [&]() -> const char* {
      if (i + 1 >= argc) usage(argv[0]);
      return argv[++i];
    };
    if (a == "-v" || a == "--verbose") verbose++;
    else if (a == "--by_index") by_index = true;
    else if (a == "--max_results") chunk_size = strtoll(next(), nullptr, 0);       // sscanf "%" SCNi64
    else if (a == "--offsets") offsets = true;
    else if (a == "--count") count = true;
    else if (a == "--matches") matches = count = true;
    else if (a == "--output") output = next();
    else if (a == "--null") sep = '\0';
    else if (a == "--json") json = true;
    else if (a == "--icase") icase = true;
    else if (a == "--literal") literal = true;
    else if (a == "--formats") return print_formats();
    else if (a == "--format-selftest") return format_selftest(sep);
    else if (a == "--device") device = atoi(next());
    else if (a == "--pattern") { pattern = next(); have_pattern = true; }
    else if (a == "--pattern-from") {
      const char* fn = next();
      if (!read_file(fn, &pattern)) { printf("Could not read pattern from %s\n", fn); return 255; }
      have_pattern = true;
    } else if (a == "--raw-pattern") { rawpattern = next(); have_raw = true; }
    else if (a == "--raw-pattern-from") {
      const char* fn = next();
      if (!read_file(fn, &rawpattern)) { printf("Could not read pattern from %s\n", fn); return 255; }
      have_raw = true;
    } else if (a == "--grep" || a == "--multigrep" || a == "--suggest") refuse(a.c_str());
    else if (a == "--grepdir" || a == "--grep-dir" || a == "--suggeststarts" || a == "--suggest-starts" || a == "--filter-results") refuse(a.c_str());
    else if (!a.empty() && a[0] == '-') {
      printf("Unknown option %s\n", a.c_str());
      usage(argv[0]);
    } else paths.push_back(a);
  }
  if (json) sep = '\n';
  if (!have_pattern && !have_raw) {
    if (paths.empty()) usage(argv[0]);
    pattern = paths.back();
    paths.pop_back();
    have_pattern = true;
  }
  if (paths.empty()) usage(argv[0]);
  if (literal && !have_raw) { rawpattern = pattern; have_raw = true; }
  if (chunk_size < 1) chunk_size = 1;

  // ---- the query (search_tool.cc:716-751)
  femto_amd_regexp_t* rx = nullptr;
  std::string text = pattern;
  if (have_raw) {          // string_node_new(rawpattern): the same tree the query \xNN\xNN... parses to
    text.clear();
    char buf[8];
    for (unsigned char c : rawpattern) { snprintf(buf, sizeof buf, "\\x%02x", c); text += buf; }
    if (rawpattern.empty()) text = "''";
  }
  // parse_string sees a C string: a pattern file stops at its first NUL byte, as strlen() stops there
  const size_t text_len = have_raw ? text.size() : strlen(text.c_str());
  int rc = femto_amd_query_compile(reinterpret_cast<const uint8_t*>(text.data()), int64_t(text_len), icase ? FEMTO_AMD_QUERY_ICASE : 0, &rx);
  if (rc) {
    fprintf(stderr, "Could not parse pattern %s\n", pattern.c_str());
    fprintf(stderr, "%s\n", femto_amd_last_error());
    return 255;
  }
  const uint16_t* lit = nullptr;
  int64_t lit_len = 0;
  const bool is_string = femto_amd_regexp_literal(rx, &lit, &lit_len) != 0;
  const femto_amd_nfa_t* nfa = femto_amd_regexp_nfa(rx);
  const char* echo = femto_amd_regexp_echo(rx);
  if (verbose) {
    if (is_string) {
      printf("Extracted pattern: ");
      fprint_alpha(stdout, std::vector<uint16_t>(lit, lit + lit_len));
      printf("\n");
    }
    printf("Query pattern is: %s\n", echo);
    printf("Chunk size is %lli\n", (long long int)chunk_size);
  }
  for (const std::string& p : paths) {
    struct stat st;
    if (stat(p.c_str(), &st) != 0) { printf("Could not open index at %s\n", p.c_str()); return 255; }
  }
  FILE* out = stdout;
  if (output && !(out = fopen(output, "w"))) {
    perror("Could not fopen");
    fprintf(stderr, "Could not open output filename '%s' for writing\n", output);
    return 255;
  }

  if (json) {
    fputs(fmt_of("json_open"), out);
    for (const char* c = echo; *c; c++) fprint_ch_json(out, (unsigned char)*c);
    fputs(fmt_of("json_results"), out);
  }
  bool first_match = true;
  std::vector<Match> all;
  for (size_t pi = 0; pi < paths.size(); pi++) {
    femto_amd_index_t* ix = nullptr;
    rc = femto_amd_open(paths[pi].c_str(), device, &ix);
    if (rc) {
      printf("Could not open index at %s\n", paths[pi].c_str());
      die("femto_amd_open", rc);
    }
    // ---- the result ranges: a string query's one range, or the automaton's sorted result list
    std::vector<int64_t> rf, rl;
    std::vector<int32_t> rlen, rcost;
    if (is_string) {
      const int32_t plen = int32_t(lit_len);
      const int64_t start = 0;
      const uint16_t dummy = 0;
      int64_t first = 0, last = -1;
      if ((rc = femto_amd_count_flat(ix, 1, &plen, lit_len ? lit : &dummy, &start, &first, &last))) die("femto_amd_count_flat", rc);
      if (last >= first) { rf.push_back(first); rl.push_back(last); rlen.push_back(plen); rcost.push_back(0); }
    } else {
      int64_t rs[2] = {0, 0}, n = 0;
      int32_t status = 0;
      rc = femto_amd_nfa_search_batch(ix, 1, nfa, 0, rs, nullptr, nullptr, nullptr, nullptr, &status, &n);
      if (rc) die("femto_amd_nfa_search_batch", rc);
      if (status) {                               // die_if_err: the reference's query failed (ERR_OVERWORKED)
        fprintf(stderr, "regular expression search failed: error %d (%s)\n", status,
                status == FEMTO_AMD_ERR_OVERWORKED ? "too much work" : "search stack full");
        return 255;
      }
      rf.resize(size_t(n) + 1); rl.resize(size_t(n) + 1); rlen.resize(size_t(n) + 1); rcost.resize(size_t(n) + 1);
      if (n && (rc = femto_amd_nfa_search_batch(ix, 1, nfa, n, rs, rf.data(), rl.data(), rlen.data(), rcost.data(), &status, &n)))
        die("femto_amd_nfa_search_batch", rc);
      rf.resize(size_t(n)); rl.resize(size_t(n)); rlen.resize(size_t(n)); rcost.resize(size_t(n));
    }
    if (count) {
      std::vector<std::vector<uint16_t>> strs;
      if (is_string) strs.assign(rf.size(), std::vector<uint16_t>(lit, lit + lit_len));
      else matched_strings(ix, rf, rlen, &strs);
      for (size_t k = 0; k < rf.size(); k++)
        if (rl[k] >= rf[k]) all.push_back(Match{rf[k], rl[k], rcost[k], strs[k], int(pi)});
    } else {
      // ---- the first result chunk: ranges in order, pieces of at most chunk_size rows, until chunk_size rows are in
      std::vector<int64_t> offs;
      int64_t got = 0;
      for (size_t k = 0; k < rf.size() && got < chunk_size; k++) {
        int64_t i = rf[k];
        while (i <= rl[k] && got < chunk_size) {
          const int64_t end = std::min(rl[k], i + chunk_size - 1);
          const size_t at = offs.size();
          offs.resize(at + size_t(end - i + 1));
          if ((rc = femto_amd_parallel_locate_range(ix, i, end, offs.data() + at))) die("femto_amd_parallel_locate_range", rc);
          got += end - i + 1;
          i = end + 1;
        }
      }
      std::vector<std::pair<int64_t, int64_t>> hits;  // (document, offset in document): resolve_location for every located row
      std::vector<int64_t> rdoc(offs.size()), roff(offs.size());
      if ((rc = femto_amd_resolve_batch(ix, int64_t(offs.size()), offs.data(), rdoc.data(), roff.data()))) die("femto_amd_resolve_batch", rc);
      hits.reserve(offs.size());
      for (size_t k = 0; k < offs.size(); k++) hits.emplace_back(rdoc[k], roff[k]);
      std::sort(hits.begin(), hits.end());             // results_create_sort_locations + unionResults: a sorted SET
      hits.erase(std::unique(hits.begin(), hits.end()), hits.end());
      DocList docs;
      int64_t prev_doc = -1;
      for (const auto& h : hits) {
        if (h.first != prev_doc) {
          const char* info = nullptr;
          int64_t len = 0;
          if ((rc = femto_amd_document_info(ix, h.first, &info, &len))) die("femto_amd_document_info", rc);
          docs.emplace_back(std::string(info, size_t(len)), std::vector<int64_t>());
          prev_doc = h.first;
        }
        docs.back().second.push_back(h.second);
      }
      print_documents(out, docs, offsets, sep, json, &first_match);
    }
    femto_amd_close(ix);
  }
  int64_t total_matches = 0;
  if (count) {
    std::stable_sort(all.begin(), all.end(), match_less);
    for (size_t i = 0; i < all.size();) {
      size_t next = i;
      int64_t num = 0;
      while (next < all.size() && all[next].match == all[i].match) {
        const Match& m = all[next];
        num += m.last - m.first + 1;
        if (verbose || by_index) fprintf(out, fmt_of("by_index_head"), paths[size_t(m.index)].c_str());
        if (verbose) fprintf(out, "For pattern %s\n", echo);
        if (verbose || by_index) {
          fprintf(out, fmt_of("by_index_row"), m.last - m.first + 1, m.first, m.last);
          fprint_alpha(out, m.match);
          fprintf(out, fmt_of("matches_row_tail"), sep);
        }
        next++;
      }
      if (json) {
        if (!first_match) fputs(fmt_of("json_row_sep"), out);
        fputs("[\"", out);
        fprint_alpha_json(out, all[i].match);
        fprintf(out, "\", %" PRIi64 "]", num);
      } else {
        print_matches_row(out, num, all[i].match, sep);
      }
      first_match = false;
      total_matches += num;
      i = next;
    }
  }
  if (json) fputs(" ]", out);
  if (count) {
    if (verbose && !json) printf("\n");
    if (json) fprintf(out, fmt_of("json_total"), total_matches);
    else print_total(out, total_matches, sep);
  }
  if (json) fputs(fmt_of("json_close"), out);
  femto_amd_regexp_free(rx);
  if (out != stdout) fclose(out);
  return 0;
}
