// regexp_nfa.hpp -- host side of regular-expression search: pattern -> Thompson NFA -> the REVERSED
// automaton the backward search simulates.
//
// The reference parses a query with a flex/bison grammar into an AST, compiles it to a Thompson NFA
// (src/main/compile_regexp.c, src/main/nfa.c) and walks the index backwards with the set of NFA states each searched
// string reaches (do_regexp_query, src/main/server.c:1656): for every (row range, state set) it asks Occ for EVERY
// character the states can read next -- a 261-way fan-out of the same leaf requests a literal search issues.  Here
// the pattern language is the byte-regular-expression part of src/main/QUERY_FORMAT.txt (literals, `.`, `[...]`
// classes, `( )`, `|`, `*`, `+`, `?`, backslash escapes, quotes; unescaped whitespace separates terms and is ignored),
// without the boolean / APPROX keywords; the fan-out runs as one batched GPU kernel per string length
// (ranges_step_kernel).  The reference's regular-expression front end cannot be built in this image (no flex/bison), so
// this path has NO reference golden vectors: it is checked against brute force over the fixture texts (the method
// of index_test.c:351-434) -- parity with the reference's result lists is UNPINNED and says so in DESIGN.md.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace femto_amd {

constexpr int kRegexAlpha = 261;          // ALPHA_SIZE: class bits are alpha codes (byte + 5)
constexpr int kRegexMaxStates = 4096;

struct CharClass {
  uint64_t w[5] = {0, 0, 0, 0, 0};
  void set(int c) { w[c >> 6] |= uint64_t(1) << (c & 63); }
  bool get(int c) const { return (w[c >> 6] >> (c & 63)) & 1u; }
  void set_byte(int b) { set(b + 5); }
  void invert_bytes() {            // complement within the 256 byte codes
    for (int b = 0; b < 256; b++) w[(b + 5) >> 6] ^= uint64_t(1) << ((b + 5) & 63);
  }
  bool empty() const { return !(w[0] | w[1] | w[2] | w[3] | w[4]); }
};

struct RegexNfa {
  // forward Thompson automaton: state s has epsilon edges eps[s] and at most one class edge (cls[s] -> to[s], to < 0: none)
  std::vector<std::vector<int>> eps;
  std::vector<CharClass> cls;
  std::vector<int> to;
  int start = 0, accept = 0;
  // reversed view: r_eps[v] = states with an epsilon edge INTO v; r_in[v] = states whose class edge leads INTO v
  std::vector<std::vector<int>> r_eps, r_in;
  int size() const { return int(to.size()); }
  int add() {
    eps.emplace_back();
    cls.emplace_back();
    to.push_back(-1);
    return size() - 1;
  }
  void finish() {
    r_eps.assign(size_t(size()), {});
    r_in.assign(size_t(size()), {});
    for (int s = 0; s < size(); s++) {
      for (int t : eps[size_t(s)]) r_eps[size_t(t)].push_back(s);
      if (to[size_t(s)] >= 0) r_in[size_t(to[size_t(s)])].push_back(s);
    }
  }
};

using StateSet = std::vector<uint64_t>;
inline bool ss_get(const StateSet& s, int i) { return (s[size_t(i) >> 6] >> (i & 63)) & 1u; }
inline void ss_set(StateSet& s, int i) { s[size_t(i) >> 6] |= uint64_t(1) << (i & 63); }

// ---- parser (recursive descent) -----------------------------------------------------------------------------------
class RegexParser {
 public:
  RegexParser(const uint8_t* p, int64_t n, RegexNfa* nfa) : p_(p), n_(n), nfa_(nfa) {}
  // returns false with *err set on a syntax error
  bool parse(std::string* err) {
    Frag f;
    if (!alt(&f)) { *err = err_; return false; }
    skip_ws();
    if (at_ < n_) { *err = "unexpected '" + std::string(1, char(p_[at_])) + "'"; return false; }
    nfa_->start = f.in;
    nfa_->accept = f.out;
    nfa_->finish();
    if (nfa_->size() > kRegexMaxStates) { *err = "regular expression too large"; return false; }
    return true;
  }

 private:
  struct Frag { int in, out; };
  const uint8_t* p_;
  int64_t n_, at_ = 0;
  RegexNfa* nfa_;
  std::string err_;
  bool fail(const std::string& m) { err_ = m; return false; }
  void skip_ws() { while (at_ < n_ && (p_[at_] == ' ' || p_[at_] == '\t' || p_[at_] == '\n' || p_[at_] == '\r')) at_++; }
  Frag lit(const CharClass& c) {
    const int a = nfa_->add(), b = nfa_->add();
    nfa_->cls[size_t(a)] = c;
    nfa_->to[size_t(a)] = b;
    return {a, b};
  }
  Frag empty() {
    const int a = nfa_->add(), b = nfa_->add();
    nfa_->eps[size_t(a)].push_back(b);
    return {a, b};
  }
  bool escape(int* byte) {   // after the backslash (QUERY_FORMAT.txt "QUOTING")
    if (at_ >= n_) return fail("dangling backslash");
    const uint8_t c = p_[at_++];
    switch (c) {
      case 'n': *byte = 0x0a; return true;
      case 't': *byte = 0x09; return true;
      case 'r': *byte = 0x0d; return true;
      case 'b': *byte = 0x08; return true;
      case 'f': *byte = 0x0c; return true;
      case 'a': *byte = 0x07; return true;
      case 'e': *byte = 0x1b; return true;
      case 'v': *byte = 0x0b; return true;
      case 'x': {
        int v = 0;
        for (int k = 0; k < 2; k++) {
          if (at_ >= n_) return fail("\\x needs two hexadecimal digits");
          const uint8_t h = p_[at_++];
          int d;
          if (h >= '0' && h <= '9') d = h - '0';
          else if (h >= 'a' && h <= 'f') d = h - 'a' + 10;
          else if (h >= 'A' && h <= 'F') d = h - 'A' + 10;
          else return fail("\\x needs two hexadecimal digits");
          v = v * 16 + d;
        }
        *byte = v;
        return true;
      }
      default: *byte = c; return true;
    }
  }
  bool char_class(Frag* out) {   // after '['
    CharClass cc;
    bool neg = false;
    if (at_ < n_ && p_[at_] == '^') { neg = true; at_++; }
    bool first = true;
    for (;;) {
      if (at_ >= n_) return fail("unterminated [");
      int lo = p_[at_++];
      if (lo == ']' && !first) break;
      if (lo == '\\' && !escape(&lo)) return false;
      int hi = lo;
      if (at_ + 1 < n_ && p_[at_] == '-' && p_[at_ + 1] != ']') {
        at_++;
        hi = p_[at_++];
        if (hi == '\\' && !escape(&hi)) return false;
        if (hi < lo) return fail("reversed range in [ ]");
      }
      for (int b = lo; b <= hi; b++) cc.set_byte(b);
      first = false;
    }
    if (neg) cc.invert_bytes();
    *out = lit(cc);
    return true;
  }
  bool quoted(uint8_t q, Frag* out) {   // "..." honours escapes, '...' is literal
    Frag f = empty();
    for (;;) {
      if (at_ >= n_) return fail("unterminated quote");
      int c = p_[at_++];
      if (c == q) break;
      if (q == '"' && c == '\\' && !escape(&c)) return false;
      CharClass cc;
      cc.set_byte(c);
      const Frag g = lit(cc);
      nfa_->eps[size_t(f.out)].push_back(g.in);
      f.out = g.out;
    }
    *out = f;
    return true;
  }
  bool atom(Frag* out) {
    skip_ws();
    if (at_ >= n_) return fail("pattern ends where a term was expected");
    const uint8_t c = p_[at_];
    if (c == '(') {
      at_++;
      if (!alt(out)) return false;
      skip_ws();
      if (at_ >= n_ || p_[at_] != ')') return fail("missing )");
      at_++;
      return true;
    }
    if (c == '[') { at_++; return char_class(out); }
    if (c == '"' || c == '\'') { at_++; return quoted(c, out); }
    if (c == '.') {
      at_++;
      CharClass cc;
      for (int b = 0; b < 256; b++) cc.set_byte(b);
      *out = lit(cc);
      return true;
    }
    if (c == ')' || c == '|' || c == '*' || c == '+' || c == '?' || c == ']' || c == '{' || c == '}')
      return fail(std::string("unexpected '") + char(c) + "'");
    at_++;
    int b = c;
    if (c == '\\' && !escape(&b)) return false;
    CharClass cc;
    cc.set_byte(b);
    *out = lit(cc);
    return true;
  }
  bool repeat(Frag* out) {
    Frag f;
    if (!atom(&f)) return false;
    for (;;) {
      skip_ws();
      if (at_ >= n_) break;
      const uint8_t c = p_[at_];
      if (c != '*' && c != '+' && c != '?') break;
      at_++;
      const int a = nfa_->add(), b = nfa_->add();
      nfa_->eps[size_t(a)].push_back(f.in);
      nfa_->eps[size_t(f.out)].push_back(b);
      if (c == '*' || c == '?') nfa_->eps[size_t(a)].push_back(b);      // zero times
      if (c == '*' || c == '+') nfa_->eps[size_t(f.out)].push_back(f.in); // again
      f = {a, b};
    }
    *out = f;
    return true;
  }
  bool concat(Frag* out) {
    Frag f = empty();
    for (;;) {
      skip_ws();
      if (at_ >= n_ || p_[at_] == '|' || p_[at_] == ')') break;
      Frag g;
      if (!repeat(&g)) return false;
      nfa_->eps[size_t(f.out)].push_back(g.in);
      f.out = g.out;
    }
    *out = f;
    return true;
  }
  bool alt(Frag* out) {
    Frag f;
    if (!concat(&f)) return false;
    skip_ws();
    while (at_ < n_ && p_[at_] == '|') {
      at_++;
      Frag g;
      if (!concat(&g)) return false;
      const int a = nfa_->add(), b = nfa_->add();
      nfa_->eps[size_t(a)].push_back(f.in);
      nfa_->eps[size_t(a)].push_back(g.in);
      nfa_->eps[size_t(f.out)].push_back(b);
      nfa_->eps[size_t(g.out)].push_back(b);
      f = {a, b};
      skip_ws();
    }
    *out = f;
    return true;
  }
};

// ---- simulation helpers ---------------------------------------------------------------------------------------------
// closure under REVERSED epsilon edges (in place)
inline void closure_rev(const RegexNfa& n, StateSet& s) {
  std::vector<int> stack;
  for (int i = 0; i < n.size(); i++) if (ss_get(s, i)) stack.push_back(i);
  while (!stack.empty()) {
    const int v = stack.back();
    stack.pop_back();
    for (int u : n.r_eps[size_t(v)])
      if (!ss_get(s, u)) { ss_set(s, u); stack.push_back(u); }
  }
}
// the states from which reading alpha code c leads INTO a state of s, closed under reversed epsilon edges
inline StateSet step_rev(const RegexNfa& n, const StateSet& s, int c) {
  StateSet r(s.size(), 0);
  for (int v = 0; v < n.size(); v++)
    if (ss_get(s, v))
      for (int u : n.r_in[size_t(v)])
        if (n.cls[size_t(u)].get(c)) ss_set(r, u);
  closure_rev(n, r);
  return r;
}
// alpha codes some state of s can be entered through (what the backward search may prepend next)
inline CharClass incoming_chars(const RegexNfa& n, const StateSet& s) {
  CharClass r;
  for (int v = 0; v < n.size(); v++)
    if (ss_get(s, v))
      for (int u : n.r_in[size_t(v)])
        for (int k = 0; k < 5; k++) r.w[k] |= n.cls[size_t(u)].w[k];
  return r;
}
// ---- approximate matching (QUERY_FORMAT.txt "APPROXIMATE SEARCH"; the error-counting states of src/main/nfa.c) -------
// A cost vector holds, per state, the least cost at which the string read so far (right to left) can have brought the
// reversed automaton there (kNoCost: not at all).  Costs: a substitution, a character missing from the data ("delete":
// the pattern's character is skipped), an extra character in the data ("insert": read without moving).
constexpr uint8_t kNoCost = 255;
struct ApproxCosts { int max_cost = 0, subst = 1, del = 1, ins = 1; };
using CostVec = std::vector<uint8_t>;

// closure: reversed epsilon edges cost nothing, skipping a class edge costs `del` (relaxed until stable; costs are small)
inline void closure_cost(const RegexNfa& n, const ApproxCosts& k, CostVec& c) {
  std::vector<int> stack;
  for (int i = 0; i < n.size(); i++) if (c[size_t(i)] != kNoCost) stack.push_back(i);
  while (!stack.empty()) {
    const int v = stack.back();
    stack.pop_back();
    const int cv = c[size_t(v)];
    for (int u : n.r_eps[size_t(v)])
      if (cv < c[size_t(u)]) { c[size_t(u)] = uint8_t(cv); stack.push_back(u); }
    if (k.max_cost > 0 && cv + k.del <= k.max_cost)
      for (int u : n.r_in[size_t(v)])
        if (cv + k.del < c[size_t(u)]) { c[size_t(u)] = uint8_t(cv + k.del); stack.push_back(u); }
  }
}
// read alpha code x; `at_end`: this is the pattern's last character -- no substitution and no extra character there
// (QUERY_FORMAT.txt: "The approximate search will never allow substitutions at the last character")
inline CostVec step_cost(const RegexNfa& n, const ApproxCosts& k, const CostVec& c, int x, bool at_end) {
  CostVec r(c.size(), kNoCost);
  for (int v = 0; v < n.size(); v++) {
    const int cv = c[size_t(v)];
    if (cv == kNoCost) continue;
    for (int u : n.r_in[size_t(v)]) {
      const int cu = n.cls[size_t(u)].get(x) ? cv : (at_end ? int(kNoCost) : cv + k.subst);
      if (cu <= k.max_cost && cu < r[size_t(u)]) r[size_t(u)] = uint8_t(cu);
    }
    if (!at_end && cv + k.ins <= k.max_cost && cv + k.ins < r[size_t(v)]) r[size_t(v)] = uint8_t(cv + k.ins);
  }
  closure_cost(n, k, r);
  return r;
}
inline bool any_alive(const CostVec& c) {
  for (uint8_t v : c) if (v != kNoCost) return true;
  return false;
}
// characters worth prepending: with errors left every character of the text can be an error; otherwise the class edges
inline CharClass incoming_chars_cost(const RegexNfa& n, const ApproxCosts& k, const CostVec& c, bool at_end, const CharClass& all) {
  CharClass r;
  bool errors_left = false;
  for (int v = 0; v < n.size(); v++) {
    const int cv = c[size_t(v)];
    if (cv == kNoCost) continue;
    if (!at_end && (cv + k.ins <= k.max_cost || (cv + k.subst <= k.max_cost && !n.r_in[size_t(v)].empty()))) errors_left = true;
    for (int u : n.r_in[size_t(v)])
      for (int w = 0; w < 5; w++) r.w[w] |= n.cls[size_t(u)].w[w];
  }
  if (errors_left) for (int w = 0; w < 5; w++) r.w[w] |= all.w[w];
  return r;
}

// does the automaton accept exactly this byte string?  (tests: the parser and the construction against a regex library)
inline bool nfa_full_match(const RegexNfa& n, const uint8_t* s, int64_t len) {
  StateSet cur(size_t(n.size() + 63) / 64, 0);
  ss_set(cur, n.accept);
  closure_rev(n, cur);
  for (int64_t i = len - 1; i >= 0; i--) cur = step_rev(n, cur, int(s[i]) + 5);   // backwards, as the index search does
  return ss_get(cur, n.start);
}

}  // namespace femto_amd
