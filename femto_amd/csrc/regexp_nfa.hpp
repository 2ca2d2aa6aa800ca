// regexp_nfa.hpp -- host side of regular-expression search: pattern -> Thompson NFA -> the epsilon-free
// automaton of the REVERSED pattern in the reference's own form (nfa_description_t, src/main/nfa.h:62-88).
//
// The reference parses a query with a flex/bison grammar into an AST, reverses it (reverse_regexp), compiles it to a
// Thompson NFA and removes the epsilon edges (src/main/compile_regexp.c:658-705); do_regexp_query (src/main/server.c:1656)
// then walks the index backwards simulating that automaton.  The pattern language is femto's (query_parser.hpp restates the
// scanner and the grammar rule by rule, without the boolean operators); the automaton handed to the search is
// the Glushkov (position) automaton of the reversed pattern -- epsilon-free by construction, one node per character
// position plus the start node.  The reference's front end cannot be generated in this image (no flex/bison), so parity is
// pinned one level below it: the SAME nfa_description_t is fed to the genuine do_regexp_query through
// setup_regexp_query_take_nfa (server.h:838; oracle/ref_tool.c regexp_nfa) and to femto_amd_nfa_search_batch
// (regexp_search.hip), and the result lists must be identical (tests/golden/*_regexp.npz, tests/test_regexp.py).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace femto_amd {

constexpr int kRegexAlpha = 261;          // ALPHA_SIZE: class bits are alpha codes (byte + 5)
constexpr int kRegexMaxStates = 4096;
constexpr int kRegexMaxDepth = 256;       // nesting of ( ): the recursive-descent parser's stack is bounded by it
constexpr int64_t kRegexMaxLen = 1 << 20; // bytes of pattern text accepted at all

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
  bool too_large = false;      // add() refuses to grow past kRegexMaxStates: the parser fails early (no multi-MB automata)
  int add() {
    if (size() >= kRegexMaxStates) { too_large = true; return size() - 1; }
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
// ---- the automaton the search simulates: the reference's nfa_description_t, flat -----------------------------------------
// Nodes with (character, destination) transitions, a set of start nodes, a set of final nodes, the approximate-search
// settings (src/main/nfa.h:62-88, regexp_settings_t src/main/index_types.h:147-162).  The search reads the matched string
// from its LAST character to its first, so this is an automaton of the reversed pattern.
struct NfaDesc {
  int32_t num_nodes = 0;
  std::vector<int32_t> trans_start;   // [num_nodes + 1]
  std::vector<int32_t> trans_char;    // alpha codes (byte + 5)
  std::vector<int32_t> trans_dest;
  std::vector<uint8_t> is_start, is_final;
  int32_t cost_bound = 1, subst_cost = 1, delete_cost = 1, insert_cost = 1;   // set_default_regexp_settings: exact matching
};
constexpr int64_t kNfaMaxTransitions = int64_t(1) << 22;

// Glushkov automaton of the reversed pattern: node 0 = nothing read yet; node 1 + k = the k-th class edge ("position") of the
// Thompson automaton has just been read.  Reading backwards, position p' may follow position p when p' can PRECEDE p in a
// match (to[p'] reaches p over epsilon edges); the first character read must be able to END a match; a node is final when
// its position can START a match.  false: more than kNfaMaxTransitions transitions.
inline bool build_reversed_nfa(const RegexNfa& n, NfaDesc* out) {
  const int S = n.size();
  std::vector<int> pos_of(size_t(S), -1), positions;
  for (int s = 0; s < S; s++)
    if (n.to[size_t(s)] >= 0) { pos_of[size_t(s)] = int(positions.size()); positions.push_back(s); }
  const int P = int(positions.size());
  // forward epsilon closure of one state
  std::vector<char> seen(static_cast<size_t>(S));
  std::vector<int> stack;
  auto closure = [&](int from) {
    std::fill(seen.begin(), seen.end(), 0);
    stack.assign(1, from);
    seen[size_t(from)] = 1;
    while (!stack.empty()) {
      const int v = stack.back();
      stack.pop_back();
      for (int u : n.eps[size_t(v)])
        if (!seen[size_t(u)]) { seen[size_t(u)] = 1; stack.push_back(u); }
    }
  };
  // preds[p] = positions p' whose edge target reaches position p; ends = positions whose edge target reaches accept
  std::vector<std::vector<int>> preds(static_cast<size_t>(P));
  std::vector<int> ends;
  for (int k = 0; k < P; k++) {
    closure(n.to[size_t(positions[size_t(k)])]);
    if (seen[size_t(n.accept)]) ends.push_back(k);
    for (int j = 0; j < P; j++)
      if (seen[size_t(positions[size_t(j)])]) preds[size_t(j)].push_back(k);
  }
  closure(n.start);
  out->num_nodes = P + 1;
  out->is_start.assign(size_t(P) + 1, 0);
  out->is_final.assign(size_t(P) + 1, 0);
  out->is_start[0] = 1;
  if (seen[size_t(n.accept)]) out->is_final[0] = 1;            // the pattern matches the empty string
  for (int k = 0; k < P; k++)
    if (seen[size_t(positions[size_t(k)])]) out->is_final[size_t(k) + 1] = 1;
  out->trans_start.assign(1, 0);
  out->trans_char.clear();
  out->trans_dest.clear();
  auto emit = [&](const std::vector<int>& targets) -> bool {
    for (int k : targets) {
      const CharClass& cc = n.cls[size_t(positions[size_t(k)])];
      for (int c = 0; c < kRegexAlpha; c++)
        if (cc.get(c)) {
          out->trans_char.push_back(c);
          out->trans_dest.push_back(k + 1);
        }
      if (int64_t(out->trans_char.size()) > kNfaMaxTransitions) return false;
    }
    out->trans_start.push_back(int32_t(out->trans_char.size()));
    return true;
  };
  if (!emit(ends)) return false;
  for (int k = 0; k < P; k++)
    if (!emit(preds[size_t(k)])) return false;
  return true;
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
