// query_parser.hpp -- femto's query language: pattern text -> syntax tree -> Thompson automaton (regexp_nfa.hpp).
//
// The reference's front end is a flex scanner (src/main/posix.flex.l) and a bison grammar (src/main/posix.bison.y) that
// build an AST (src/main/ast.c); femto_search then rewrites it with streamline_query (src/main/query_planning.c:24-218),
// turns a query without alternatives into a plain string (simplify_query / get_simple_query, ast.c:1155-1269) and, with
// --icase, widens every character to both cases (icase_ast, ast.c:457-589) before compile_regexp.c makes the automaton.
// flex and bison are not in this image, so none of that can be generated; this file restates the TOKEN RULES and the
// GRAMMAR by hand, rule by rule (each function names the rule it follows), so that femto_amd_search accepts what
// femto_search accepts and means the same by it -- including the corners:
//   * unescaped whitespace separates terms and is dropped; `#` starts a comment;
//   * three or more letters/digits/bytes >= 0x80 in a row form ONE string token, except that the last one is split off
//     when punctuation follows (posix.flex.l:285-318: "abcd*" is abc d*, but "abcd *" is (abcd)*);
//   * a repeat operator applies to the whole preceding token, so "'ab'+" and "{x 41 42}+" repeat two bytes;
//   * an atom takes ONE repeat operator ("a**" is a syntax error), a sequence is never empty ("a|" and "()" are errors);
//   * "{" that starts neither {x hex} nor {m}, {m,}, {m,n} is an ordinary character, as are "]", "}", "-", "," outside [ ];
//   * "\x4" (one hex digit) is the characters x and 4; "\x-02" is the alpha code 5 - 2 (end-of-document marker);
//   * APPROX is only a keyword when its argument is followed by whitespace and -- without an argument -- when TWO
//     whitespace characters follow it (the scanner rule's trailing context, posix.flex.l:277): "APPROX black" searches for
//     the string "APPROXblack";
//   * streamline_query drops leading and trailing optional parts and trims leading/trailing repeats to their minimum
//     ("a*(bc|d)+" is searched as "(bc|d)"), skipping the FIRST alternative of a trailing group (the loop at
//     query_planning.c:177 stops at i > 0) -- pinned by the known answers of src/main/query_planning_test.c.
// Boolean queries (AND OR NOT THEN WITHIN, document-level result sets: SURVEY.md 8 "out of scope") are recognised and
// refused by name.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "regexp_nfa.hpp"

namespace femto_amd {

constexpr int kUnbounded = 0x7fffffff;    // UNBOUNDED_REPEATS (ast.h:114)
constexpr int kMaxRepeat = 4096;          // a larger {m,n} cannot fit kRegexMaxStates anyway: refused at parse time

// ---- syntax tree (ast.h: regexp_node / sequence_node / atom_node / set_node / character_node / string_node) -------------
struct QRegexp;
struct QAtom {
  enum Kind { CHARACTER, SET, STRING, GROUP } kind = CHARACTER;
  int ch = 0;                      // CHARACTER: alpha code
  CharClass set;                   // SET
  std::vector<uint16_t> str;       // STRING: alpha codes
  std::vector<QRegexp> group;      // GROUP: exactly one element (vector: QRegexp is incomplete here)
  int rmin = 1, rmax = 1;          // atom_node.repeat
};
struct QSequence { std::vector<QAtom> atoms; };
struct QRegexp {
  std::vector<QSequence> choices;
  int cost_bound = 1, subst_cost = 1, delete_cost = 1, insert_cost = 1;   // regexp_settings_t (set_default_regexp_settings)
  bool has_approx = false;           // the text itself began with APPROX (an explicit "APPROX 0" is a setting too: cost_bound 1)
};

inline bool q_is_space(int c) { return c == ' ' || (c >= '\t' && c <= '\r'); }                       // [[:space:]], C locale
inline bool q_is_punct(int c) { return (c >= 33 && c <= 47) || (c >= 58 && c <= 64) || (c >= 91 && c <= 96) || (c >= 123 && c <= 126); }
inline bool q_is_word(int c) { return !q_is_space(c) && !q_is_punct(c); }                             // [^[:space:][:punct:]]
inline bool q_is_digit(int c) { return c >= '0' && c <= '9'; }
inline int q_hex(int c) { return c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1; }
inline int q_escape(int x) {       // handle_escape (posix.flex.l:113-127)
  switch (x) {
    case 'n': return 0x0a;
    case 't': return 0x09;
    case 'r': return 0x0d;
    case 'b': return 0x08;
    case 'f': return 0x0c;
    case 'a': return 0x07;
    case 'e': return 0x1b;
    case 'v': return 0x0b;
    default: return x;
  }
}

// ---- tokens (posix.flex.l) ----------------------------------------------------------------------------------------------
struct QTok {
  enum Kind { END, CHARACTER, ANY_PERIOD, STRING, REPEAT_RANGE, APPROX, SET_START, NEGATED_SET_START, SET_END, SET_DASH,
              GROUP_START, GROUP_END, REPEAT_ANY, REPEAT_PLUS, REPEAT_QUESTION, OR, BOOL } kind = END;
  int ch = 0;                     // CHARACTER: alpha code (may be < 5 for \x-NN, or negative: refused by the parser)
  std::vector<uint16_t> str;      // STRING
  int rmin = 0, rmax = 0;         // REPEAT_RANGE
  int approx[4] = {1, 1, 1, 1};   // APPROX: cost_bound, subst, delete, insert
  const char* word = "";          // BOOL: which keyword
  int64_t at = 0;                 // byte offset in the pattern (error messages)
};

class QueryLexer {
 public:
  QueryLexer(const uint8_t* p, int64_t n) : p_(p), n_(n) {}
  bool run(std::vector<QTok>* out, std::string* err) {
    bool bracket = false;
    int64_t i = 0;
    while (i < n_) {
      QTok t;
      t.at = i;
      const int c = p_[i];
      if (bracket) {                                   // <bracket> rules (posix.flex.l:214-238)
        if (c == ']') { t.kind = QTok::SET_END; bracket = false; i++; }
        else if (c == '-') { t.kind = QTok::SET_DASH; i++; }
        else if (c == '\\') {
          if (i + 3 < n_ && p_[i + 1] == 'x' && q_hex(p_[i + 2]) >= 0 && q_hex(p_[i + 3]) >= 0) {
            t.kind = QTok::CHARACTER; t.ch = 5 + q_hex(p_[i + 2]) * 16 + q_hex(p_[i + 3]); i += 4;
          } else if (i + 1 < n_) { t.kind = QTok::CHARACTER; t.ch = 5 + q_escape(p_[i + 1]); i += 2; }
          else { i++; continue; }                      // a lone backslash at the end: no rule, the scanner's default echoes it
        } else { t.kind = QTok::CHARACTER; t.ch = 5 + c; i++; }   // whitespace is literal inside [ ]
        out->push_back(t);
        continue;
      }
      if (q_is_space(c)) { i++; continue; }
      if (c == '#') { while (i < n_ && p_[i] != '\n') i++; continue; }          // comment
      if (c == '\'') {                                 // single-quoted: everything up to the next ' is literal
        int64_t j = i + 1;
        while (j < n_ && p_[j] != '\'') j++;
        if (j >= n_) { *err = "unterminated ' quote"; return false; }
        t.kind = QTok::STRING;
        for (int64_t k = i + 1; k < j; k++) t.str.push_back(uint16_t(5 + p_[k]));
        i = j + 1;
      } else if (c == '"') {                           // double-quoted: \xNN and \c escapes (posix.flex.l:166-186)
        int64_t j = i + 1;
        t.kind = QTok::STRING;
        for (;;) {
          if (j >= n_) { *err = "unterminated \" quote"; return false; }
          const int d = p_[j];
          if (d == '"') break;
          if (d == '\\') {
            if (j + 3 < n_ && p_[j + 1] == 'x' && q_hex(p_[j + 2]) >= 0 && q_hex(p_[j + 3]) >= 0) {
              t.str.push_back(uint16_t(5 + q_hex(p_[j + 2]) * 16 + q_hex(p_[j + 3])));
              j += 4;
            } else if (j + 1 < n_) { t.str.push_back(uint16_t(5 + q_escape(p_[j + 1]))); j += 2; }
            else { *err = "unterminated \" quote"; return false; }
          } else { t.str.push_back(uint16_t(5 + d)); j++; }
        }
        i = j + 1;
      } else if (c == '\\') {
        if (i + 4 < n_ && p_[i + 1] == 'x' && p_[i + 2] == '-' && q_hex(p_[i + 3]) >= 0 && q_hex(p_[i + 4]) >= 0) {
          t.kind = QTok::CHARACTER; t.ch = 5 - (q_hex(p_[i + 3]) * 16 + q_hex(p_[i + 4])); i += 5;     // "\x-"NN: headers, EOF
        } else if (i + 3 < n_ && p_[i + 1] == 'x' && q_hex(p_[i + 2]) >= 0 && q_hex(p_[i + 3]) >= 0) {
          t.kind = QTok::CHARACTER; t.ch = 5 + q_hex(p_[i + 2]) * 16 + q_hex(p_[i + 3]); i += 4;
        } else if (i + 1 < n_) { t.kind = QTok::CHARACTER; t.ch = 5 + q_escape(p_[i + 1]); i += 2; }
        else { t.kind = QTok::CHARACTER; t.ch = 5 + '\\'; i++; }                 // catch-all rule
      } else if (c == '[') {
        if (i + 1 < n_ && p_[i + 1] == '^') { t.kind = QTok::NEGATED_SET_START; i += 2; }
        else { t.kind = QTok::SET_START; i++; }
        bracket = true;
      } else if (c == '(') { t.kind = QTok::GROUP_START; i++; }
      else if (c == ')') { t.kind = QTok::GROUP_END; i++; }
      else if (c == '|') { t.kind = QTok::OR; i++; }
      else if (c == '*') { t.kind = QTok::REPEAT_ANY; i++; }
      else if (c == '+') { t.kind = QTok::REPEAT_PLUS; i++; }
      else if (c == '?') { t.kind = QTok::REPEAT_QUESTION; i++; }
      else if (c == '.') { t.kind = QTok::ANY_PERIOD; i++; }
      else if (c == '{' && brace(i, &t, &i)) { /* {x hex} or {m,n} */ }
      else if (q_is_word(c) && keyword(i, &t, &i)) { /* AND OR NOT THEN WITHIN APPROX */ }
      else if (q_is_word(c) && i + 2 < n_ && q_is_word(p_[i + 1]) && q_is_word(p_[i + 2])) {
        // <word> (posix.flex.l:285-318): the run of word characters; if punctuation follows, its last character stays behind
        int64_t e = i + 1;
        while (e < n_ && q_is_word(p_[e])) e++;
        const bool punct_next = e < n_ && q_is_punct(p_[e]);
        const int64_t end = punct_next ? e - 1 : e;
        t.kind = QTok::STRING;
        for (int64_t k = i; k < end; k++) t.str.push_back(uint16_t(5 + p_[k]));
        i = punct_next ? e - 1 : (e < n_ ? e + 1 : e);         // the terminating whitespace is consumed with the word
      } else { t.kind = QTok::CHARACTER; t.ch = 5 + c; i++; }  // catch-all [^[:space:]]
      out->push_back(t);
    }
    if (bracket) { *err = "unterminated ["; return false; }
    QTok e;
    e.at = n_;
    out->push_back(e);
    return true;
  }

 private:
  const uint8_t* p_;
  int64_t n_;
  static int scan_i(const std::string& s) { return int(strtol(s.c_str(), nullptr, 0)); }      // sscanf("%i"): 0x / 0 prefixes count
  // "{x"[[:xdigit:][:space:]]*"}" and "{"[[:digit:]]+,?[[:digit:]]*"}" (posix.flex.l:263-270)
  bool brace(int64_t i, QTok* t, int64_t* next) {
    if (i + 1 >= n_) return false;
    if (p_[i + 1] == 'x') {
      int64_t j = i + 2;
      while (j < n_ && (q_hex(p_[j]) >= 0 || q_is_space(p_[j]))) j++;
      if (j >= n_ || p_[j] != '}') return false;
      // construct_hex_string (ast.c:41-88): EVERY hex digit of the token text counts -- the `x` of "{x" is not one, but
      // the scan starts at the brace, so nothing else sneaks in; an odd digit at the end is dropped
      std::vector<int> digits;
      for (int64_t k = i; k <= j; k++) if (q_hex(p_[k]) >= 0) digits.push_back(q_hex(p_[k]));
      t->kind = QTok::STRING;
      for (size_t k = 0; k + 1 < digits.size(); k += 2) t->str.push_back(uint16_t(5 + digits[k] * 16 + digits[k + 1]));
      *next = j + 1;
      return true;
    }
    if (!q_is_digit(p_[i + 1])) return false;
    int64_t j = i + 1;
    while (j < n_ && q_is_digit(p_[j])) j++;
    const std::string a(reinterpret_cast<const char*>(p_ + i + 1), size_t(j - i - 1));
    std::string b;
    bool comma = false;
    if (j < n_ && p_[j] == ',') {
      comma = true;
      j++;
      const int64_t s = j;
      while (j < n_ && q_is_digit(p_[j])) j++;
      b.assign(reinterpret_cast<const char*>(p_ + s), size_t(j - s));
    }
    if (j >= n_ || p_[j] != '}') return false;
    t->kind = QTok::REPEAT_RANGE;                     // construct_range (posix.flex.l:56-104)
    t->rmin = scan_i(a);
    t->rmax = !comma ? t->rmin : b.empty() ? kUnbounded : scan_i(b);
    *next = j + 1;
    return true;
  }
  bool word_at(int64_t i, const char* up, const char* lo, int len) const {
    if (i + len > n_) return false;
    bool u = true, l = true;
    for (int k = 0; k < len; k++) {
      u = u && p_[i + k] == uint8_t(up[k]);
      l = l && p_[i + k] == uint8_t(lo[k]);
    }
    return u || l;
  }
  // the keyword rules with their trailing contexts (posix.flex.l:248-280)
  bool keyword(int64_t i, QTok* t, int64_t* next) {
    auto space_at = [&](int64_t k) { return k < n_ && q_is_space(p_[k]); };
    struct KW { const char* up; const char* lo; int len; };
    static const KW plain[] = {{"AND", "and", 3}, {"OR", "or", 2}, {"NOT", "not", 3}, {"THEN", "then", 4}};
    for (const KW& k : plain)
      if (word_at(i, k.up, k.lo, k.len) && space_at(i + k.len)) {
        t->kind = QTok::BOOL;
        t->word = k.up;
        *next = i + k.len;
        return true;
      }
    if (word_at(i, "WITHIN", "within", 6) && space_at(i + 6)) {      // [[:space:]]+[[:digit:]]+/[[:space:]]
      int64_t j = i + 6;
      while (space_at(j)) j++;
      const int64_t d = j;
      while (j < n_ && q_is_digit(p_[j])) j++;
      if (j > d && space_at(j)) { t->kind = QTok::BOOL; t->word = "WITHIN"; *next = j; return true; }
      return false;
    }
    if (word_at(i, "APPROX", "approx", 6) && space_at(i + 6)) {
      // ("APPROX"|"approx")[[:space:]]+([[:digit:]]+(:[[:digit:]]+){0,3})?/[[:space:]]
      int64_t j = i + 6;
      while (space_at(j)) j++;
      const int64_t spaces_end = j;
      int64_t best = -1;
      if (j < n_ && q_is_digit(p_[j])) {
        int64_t k = j;
        while (k < n_ && q_is_digit(p_[k])) k++;
        if (space_at(k)) best = k;
        for (int g = 0; g < 3 && k + 1 < n_ && p_[k] == ':' && q_is_digit(p_[k + 1]); g++) {
          k++;
          while (k < n_ && q_is_digit(p_[k])) k++;
          if (space_at(k)) best = k;
        }
      }
      if (best < 0 && spaces_end - (i + 6) >= 2) best = spaces_end - 1;     // no argument: the last space is the context
      if (best < 0) return false;
      // approx_node_new (ast.c:167-192): one error by default, arguments max_cost[:subst[:delete[:insert]]]
      int v[4] = {1, 1, 1, 1};
      v[0] = 1;                                         // cost_bound 1, ++, -- : the default argument is "1 error"
      {
        int64_t k = i + 6;
        while (space_at(k)) k++;
        for (int f = 0; f < 4 && k < best && q_is_digit(p_[k]); f++) {
          const int64_t s = k;
          while (k < best && q_is_digit(p_[k])) k++;
          v[f] = scan_i(std::string(reinterpret_cast<const char*>(p_ + s), size_t(k - s)));
          if (k < best && p_[k] == ':') k++; else break;
        }
      }
      t->kind = QTok::APPROX;
      t->approx[0] = v[0] + 1;
      t->approx[1] = v[1];
      t->approx[2] = v[2];
      t->approx[3] = v[3];
      *next = best;
      return true;
    }
    return false;
  }
};

// ---- grammar (posix.bison.y) ------------------------------------------------------------------------------------------
class QueryParser {
 public:
  explicit QueryParser(const std::vector<QTok>& toks) : t_(toks) {}
  bool parse(QRegexp* out, std::string* err) {
    for (const QTok& t : t_)
      if (t.kind == QTok::BOOL) {
        *err = std::string("boolean queries are not supported (") + t.word + " at byte " + std::to_string(t.at) +
               "): this tool searches byte patterns; quote the word to search for it";
        return false;
      }
    bool approx = false;
    QTok ap;
    if (cur().kind == QTok::APPROX) { approx = true; ap = cur(); k_++; }          // regexp_top: T_APPROX regexp
    if (!regexp(out)) { *err = err_; return false; }
    if (cur().kind != QTok::END) { *err = "syntax error at byte " + std::to_string(cur().at); return false; }
    if (approx) {
      out->has_approx = true;
      out->cost_bound = ap.approx[0];
      out->subst_cost = ap.approx[1];
      out->delete_cost = ap.approx[2];
      out->insert_cost = ap.approx[3];
    }
    return true;
  }

 private:
  const std::vector<QTok>& t_;
  size_t k_ = 0;
  int depth_ = 0;
  std::string err_;
  const QTok& cur() const { return t_[k_]; }
  bool fail(const std::string& m) { err_ = m + " at byte " + std::to_string(cur().at); return false; }
  bool starts_atom() const {
    switch (cur().kind) {
      case QTok::GROUP_START: case QTok::SET_START: case QTok::NEGATED_SET_START: case QTok::CHARACTER: case QTok::ANY_PERIOD:
      case QTok::STRING: return true;
      default: return false;
    }
  }
  bool regexp(QRegexp* r) {                       // regexp: sequence | regexp T_OR sequence
    for (;;) {
      QSequence s;
      if (!sequence(&s)) return false;
      r->choices.push_back(std::move(s));
      if (cur().kind != QTok::OR) return true;
      k_++;
    }
  }
  bool sequence(QSequence* s) {                   // sequence: piece | sequence piece   (never empty)
    if (!starts_atom()) return fail(cur().kind == QTok::END ? "pattern ends where a term was expected" : "syntax error");
    while (starts_atom()) {
      QAtom a;
      if (!piece(&a)) return false;
      s->atoms.push_back(std::move(a));
    }
    return true;
  }
  bool piece(QAtom* a) {                          // piece: atom | atom repeat_op   (one operator)
    if (!atom(a)) return false;
    switch (cur().kind) {
      case QTok::REPEAT_ANY: a->rmin = 0; a->rmax = kUnbounded; k_++; break;
      case QTok::REPEAT_PLUS: a->rmin = 1; a->rmax = kUnbounded; k_++; break;
      case QTok::REPEAT_QUESTION: a->rmin = 0; a->rmax = 1; k_++; break;
      case QTok::REPEAT_RANGE:
        a->rmin = cur().rmin;
        a->rmax = cur().rmax;
        if (a->rmin > kMaxRepeat || (a->rmax != kUnbounded && a->rmax > kMaxRepeat)) return fail("repeat count too large");
        k_++;
        break;
      default: break;
    }
    return true;
  }
  bool atom(QAtom* a) {
    const QTok& t = cur();
    switch (t.kind) {
      case QTok::GROUP_START: {
        k_++;
        if (++depth_ > kRegexMaxDepth) return fail("parentheses nested too deeply");
        QRegexp r;
        const bool ok = regexp(&r);
        depth_--;
        if (!ok) return false;
        if (cur().kind != QTok::GROUP_END) return fail("missing )");
        k_++;
        a->kind = QAtom::GROUP;
        a->group.push_back(std::move(r));
        return true;
      }
      case QTok::SET_START: case QTok::NEGATED_SET_START: {
        const bool neg = t.kind == QTok::NEGATED_SET_START;
        k_++;
        a->kind = QAtom::SET;
        int items = 0;
        while (cur().kind == QTok::CHARACTER) {            // bracket_item: T_CHARACTER T_SET_DASH T_CHARACTER | T_CHARACTER
          const int lo = cur().ch;
          int hi = lo;
          k_++;
          if (cur().kind == QTok::SET_DASH) {
            k_++;
            if (cur().kind != QTok::CHARACTER) return fail("syntax error in [ ]");
            hi = cur().ch;
            k_++;
          }
          for (int c = lo; c <= hi; c++)                    // set_node_set: codes outside 0..260 are ignored; a reversed range is empty
            if (c >= 0 && c < kRegexAlpha) a->set.set(c);
          items++;
        }
        if (cur().kind != QTok::SET_END || !items) return fail("syntax error in [ ]");
        k_++;
        if (neg) {                                          // set_node_invert (ast.c:324-340): complement within the 256 bytes
          CharClass inv;
          for (int b = 0; b < 256; b++) if (!a->set.get(b + 5)) inv.set(b + 5);
          a->set = inv;
        }
        return true;
      }
      case QTok::CHARACTER:
        if (t.ch < 0 || t.ch >= kRegexAlpha) return fail("character outside the alphabet");
        a->kind = QAtom::CHARACTER;
        a->ch = t.ch;
        k_++;
        return true;
      case QTok::ANY_PERIOD:                                // period_range (ast.c:33): the 256 bytes
        a->kind = QAtom::SET;
        for (int b = 0; b < 256; b++) a->set.set(b + 5);
        k_++;
        return true;
      case QTok::STRING:
        a->kind = QAtom::STRING;
        a->str = t.str;
        k_++;
        return true;
      default: return fail("syntax error");
    }
  }
};

// ---- streamline_query (query_planning.c:24-218) -------------------------------------------------------------------------
inline bool q_matches_empty(const QRegexp& r);
inline bool q_matches_empty(const QAtom& a) {              // matches_empty_string, AST_NODE_ATOM
  if (a.rmin == 0) return true;
  if (a.kind == QAtom::GROUP) return q_matches_empty(a.group[0]);
  return false;
}
inline bool q_matches_empty(const QSequence& s) {
  for (const QAtom& a : s.atoms) if (!q_matches_empty(a)) return false;
  return true;
}
inline bool q_matches_empty(const QRegexp& r) {
  for (const QSequence& s : r.choices) if (q_matches_empty(s)) return true;
  return false;
}
inline void q_fix_initial(QRegexp& r);
inline void q_fix_initial(QSequence& s) {
  while (!s.atoms.empty() && q_matches_empty(s.atoms.front())) s.atoms.erase(s.atoms.begin());
  if (s.atoms.empty()) return;
  QAtom& a = s.atoms.front();
  if (a.rmax > a.rmin) a.rmax = a.rmin;
  if (a.kind == QAtom::GROUP) q_fix_initial(a.group[0]);
}
inline void q_fix_initial(QRegexp& r) { for (QSequence& s : r.choices) q_fix_initial(s); }
inline void q_fix_final(QRegexp& r);
inline void q_fix_final(QSequence& s) {
  while (!s.atoms.empty() && q_matches_empty(s.atoms.back())) s.atoms.pop_back();
  if (s.atoms.empty()) return;
  QAtom& a = s.atoms.back();
  if (a.rmax > a.rmin) a.rmax = a.rmin;
  if (a.kind == QAtom::GROUP) q_fix_final(a.group[0]);
}
inline void q_fix_final(QRegexp& r) {                      // "for (i = num - 1; i > 0; i--)": the first alternative is left alone
  for (size_t i = r.choices.size(); i-- > 1;) q_fix_final(r.choices[i]);
}
inline void q_streamline(QRegexp& r) {                     // the top level visits every alternative, nested groups go through the loops above
  for (QSequence& s : r.choices) {
    q_fix_initial(s);
    q_fix_final(s);
  }
}

// ---- simplify_query / get_simple_query (ast.c:1155-1269): a query that is one string ----------------------------------------
inline bool q_simple(const QRegexp& r, std::vector<uint16_t>* out);
inline bool q_simple(const QAtom& a, std::vector<uint16_t>* out) {
  if (a.rmin != a.rmax) return false;
  for (int k = 0; k < a.rmin; k++) {
    switch (a.kind) {
      case QAtom::CHARACTER: out->push_back(uint16_t(a.ch)); break;
      case QAtom::STRING: out->insert(out->end(), a.str.begin(), a.str.end()); break;
      case QAtom::SET: {
        int n = 0, last = 0;
        for (int c = 0; c < kRegexAlpha; c++) if (a.set.get(c)) { n++; last = c; }
        if (n != 1) return false;
        out->push_back(uint16_t(last));
        break;
      }
      case QAtom::GROUP:
        if (!q_simple(a.group[0], out)) return false;
        break;
    }
    if (out->size() > size_t(kRegexMaxLen)) return false;
  }
  return true;
}
inline bool q_simple(const QRegexp& r, std::vector<uint16_t>* out) {
  if (r.choices.size() != 1 || r.cost_bound > 1) return false;
  for (const QAtom& a : r.choices[0].atoms) if (!q_simple(a, out)) return false;
  return true;
}

// ---- icase_ast (ast.c:457-589; toloweralpha / toupperalpha index_types.h:74-83: C-locale tolower/toupper of the byte) ----
// (A code below the bytes -- \x-NN -- goes through tolower() as a NEGATIVE int; glibc's table answers -128..-1 like the byte
// 256 + c, so the reference's --icase turns alpha 2 into alpha 258 = byte 0xfd.  Kept: tests/golden/query_ast_golden.json.)
inline int q_lower(int alpha) { const int b = alpha - 5; return b < 0 ? alpha + 256 : (b >= 'A' && b <= 'Z' ? alpha + 32 : alpha); }
inline int q_upper(int alpha) { const int b = alpha - 5; return b < 0 ? alpha + 256 : (b >= 'a' && b <= 'z' ? alpha - 32 : alpha); }
inline CharClass q_both_cases(int alpha) {
  CharClass c;
  c.set(q_lower(alpha));
  c.set(q_upper(alpha));
  return c;
}
inline void q_icase(QRegexp& r) {
  for (QSequence& s : r.choices) {
    std::vector<QAtom> atoms;
    for (QAtom& a : s.atoms) {
      switch (a.kind) {
        case QAtom::CHARACTER: a.kind = QAtom::SET; a.set = q_both_cases(a.ch); atoms.push_back(std::move(a)); break;
        case QAtom::SET: {
          CharClass c = a.set;
          for (int k = 0; k < kRegexAlpha; k++) if (a.set.get(k)) { c.set(q_lower(k)); c.set(q_upper(k)); }
          a.set = c;
          atoms.push_back(std::move(a));
          break;
        }
        case QAtom::STRING: {                    // a string becomes a sequence of two-character sets, repeated as a group
          QAtom g;
          g.kind = QAtom::GROUP;
          g.rmin = a.rmin;
          g.rmax = a.rmax;
          QRegexp inner;
          inner.choices.emplace_back();
          for (uint16_t ch : a.str) {
            QAtom x;
            x.kind = QAtom::SET;
            x.set = q_both_cases(ch);
            inner.choices[0].atoms.push_back(std::move(x));
          }
          g.group.push_back(std::move(inner));
          atoms.push_back(std::move(g));
          break;
        }
        case QAtom::GROUP: q_icase(a.group[0]); atoms.push_back(std::move(a)); break;
      }
    }
    s.atoms.swap(atoms);
  }
}

// ---- ast_to_string (ast.c:875-1120): the query echoed back --------------------------------------------------------------------
enum QCtx { Q_IN_RE, Q_IN_SET, Q_IN_DQUOTES };
inline void q_char_append(std::string& o, int alpha, QCtx ctx) {       // ast_char_append
  static const char* esc_re = "[]()|*+?-{}.'\"\\";
  static const char* esc_set = "]-\\";
  const int chr = alpha - 5;
  auto in = [](const char* s, int c) { for (; *s; s++) if (*s == c) return true; return false; };
  char buf[16];
  if ((ctx == Q_IN_DQUOTES && chr == '"') || (ctx == Q_IN_SET && chr > 0 && in(esc_set, chr)) || (ctx == Q_IN_RE && chr > 0 && in(esc_re, chr))) {
    o.push_back('\\');
    o.push_back(char(chr));
  } else if (chr > 0 && ((chr > 32 && chr < 127) || chr == ' ')) {
    o.push_back(char(chr));
  } else if (chr < 0) {
    snprintf(buf, sizeof buf, "\\x-%02x", -chr);
    o += buf;
  } else {
    snprintf(buf, sizeof buf, "\\x%02x", chr);
    o += buf;
  }
}
inline void q_echo(const QRegexp& r, std::string& o, bool usequotes);
inline void q_echo_atom(const QAtom& a, std::string& o, bool usequotes) {
  bool justone = a.rmin == 1 && a.rmax == 1;
  if (a.kind == QAtom::CHARACTER) justone = true;
  if (!justone) o.push_back('(');
  switch (a.kind) {
    case QAtom::CHARACTER: q_char_append(o, a.ch, Q_IN_RE); break;
    case QAtom::STRING:
      if (usequotes) {
        if (!o.empty() && o.back() != ' ') o.push_back(' ');
        o.push_back('"');
      }
      for (uint16_t c : a.str) q_char_append(o, c, usequotes ? Q_IN_DQUOTES : Q_IN_RE);
      if (usequotes) o.push_back('"');
      break;
    case QAtom::GROUP: q_echo(a.group[0], o, usequotes); break;
    case QAtom::SET: {
      int nset = 0, last = 0;
      bool period = true;
      for (int c = 0; c < kRegexAlpha; c++) {
        const bool in_period = c >= 5 && c <= 260;
        if (a.set.get(c)) { nset++; last = c; if (!in_period) period = false; }
        else if (in_period) period = false;
      }
      if (period) o.push_back('.');
      else if (nset == 1) q_char_append(o, last, Q_IN_SET);
      else {
        o.push_back('[');
        for (int s = 0; s < kRegexAlpha;) {
          int e = s + 1;
          if (a.set.get(s)) {
            while (e < kRegexAlpha && a.set.get(e)) e++;
            if (e - s == 1) q_char_append(o, s, Q_IN_SET);
            else if (e - s == 2) { q_char_append(o, s, Q_IN_SET); q_char_append(o, s + 1, Q_IN_SET); }
            else { q_char_append(o, s, Q_IN_SET); o.push_back('-'); q_char_append(o, e - 1, Q_IN_SET); }
          }
          s = e;
        }
        o.push_back(']');
      }
      break;
    }
  }
  if (!justone) o.push_back(')');
  char buf[40];
  if (a.rmax == kUnbounded) {
    if (a.rmin == 0) o.push_back('*');
    else if (a.rmin == 1) o.push_back('+');
    else { snprintf(buf, sizeof buf, "{%i,}", a.rmin); o += buf; }
  } else if (a.rmin == 1 && a.rmax == 1) {
  } else if (a.rmin == 0 && a.rmax == 1) o.push_back('?');
  else if (a.rmin == a.rmax) { snprintf(buf, sizeof buf, "{%i}", a.rmin); o += buf; }
  else { snprintf(buf, sizeof buf, "{%i,%i}", a.rmin, a.rmax); o += buf; }
}
inline void q_echo(const QRegexp& r, std::string& o, bool usequotes) {
  if (r.choices.size() > 1) o.push_back('(');
  for (size_t i = 0; i < r.choices.size(); i++) {
    for (const QAtom& a : r.choices[i].atoms) q_echo_atom(a, o, usequotes);
    if (r.choices.size() > 1 && i + 1 < r.choices.size()) o.push_back('|');
  }
  if (r.choices.size() > 1) o.push_back(')');
}

// ---- the parsed tree as text, for oracle/ref_tool.c `ast` (tests/golden/make_query_golden.py): the genuine reference rebuilds
// it with its own constructors (ast.h) and runs its own streamline_query / simplify_query / icase_ast / ast_to_string on it --
// everything behind the generated parser is then pinned to the reference, not to a restatement.  Format, blank-separated:
//   R <cost_bound> <subst> <delete> <insert> <nchoices> { S <natoms> { A <rmin> <rmax> <kind ...> } }
//   kinds: C <alpha> | T <n> <alpha>... (set) | G <n> <alpha>... (string) | P R ... (group)
inline void q_dump(const QRegexp& r, std::string& o) {
  o += "R " + std::to_string(r.cost_bound) + " " + std::to_string(r.subst_cost) + " " + std::to_string(r.delete_cost) + " " +
       std::to_string(r.insert_cost) + " " + std::to_string(r.choices.size()) + " ";
  for (const QSequence& s : r.choices) {
    o += "S " + std::to_string(s.atoms.size()) + " ";
    for (const QAtom& a : s.atoms) {
      o += "A " + std::to_string(a.rmin) + " " + std::to_string(a.rmax) + " ";
      switch (a.kind) {
        case QAtom::CHARACTER: o += "C " + std::to_string(a.ch) + " "; break;
        case QAtom::SET: {
          int n = 0;
          for (int c = 0; c < kRegexAlpha; c++) n += a.set.get(c) ? 1 : 0;
          o += "T " + std::to_string(n) + " ";
          for (int c = 0; c < kRegexAlpha; c++) if (a.set.get(c)) o += std::to_string(c) + " ";
          break;
        }
        case QAtom::STRING:
          o += "G " + std::to_string(a.str.size()) + " ";
          for (uint16_t c : a.str) o += std::to_string(c) + " ";
          break;
        case QAtom::GROUP: o += "P "; q_dump(a.group[0], o); break;
      }
    }
  }
}

// simplify_query (ast.c:1239-1269): a query that is one string is REPLACED by a string node
inline void q_simplify(QRegexp& q) {
  std::vector<uint16_t> lit;
  if (!q_simple(q, &lit)) return;
  QAtom a;
  a.kind = QAtom::STRING;
  a.str.swap(lit);
  QSequence seq;
  seq.atoms.push_back(std::move(a));
  q.choices.clear();
  q.choices.push_back(std::move(seq));
}

// ---- syntax tree -> Thompson automaton (compile_regexp_thompson, compile_regexp.c:150-360: repeats are copies) -------------
class QueryCompiler {
 public:
  explicit QueryCompiler(RegexNfa* nfa) : nfa_(nfa) {}
  bool compile(const QRegexp& r, std::string* err) {
    const Frag f = regexp(r);
    if (nfa_->too_large) { *err = "regular expression too large"; return false; }
    nfa_->start = f.in;
    nfa_->accept = f.out;
    nfa_->finish();
    return true;
  }

 private:
  struct Frag { int in, out; };
  RegexNfa* nfa_;
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
  void link(Frag& f, const Frag& g) {
    nfa_->eps[size_t(f.out)].push_back(g.in);
    f.out = g.out;
  }
  Frag base(const QAtom& a) {
    switch (a.kind) {
      case QAtom::CHARACTER: { CharClass c; c.set(a.ch); return lit(c); }
      case QAtom::SET: return lit(a.set);
      case QAtom::STRING: {
        Frag f = empty();
        for (uint16_t ch : a.str) {
          if (nfa_->too_large) break;
          CharClass c;
          c.set(ch);
          link(f, lit(c));
        }
        return f;
      }
      case QAtom::GROUP: return regexp(a.group[0]);
    }
    return empty();
  }
  // x{m,n}: m copies, then n - m optional ones; x{m,}: the LAST mandatory copy loops (x{0,}: one optional looping copy).
  // compile_regexp.c:222-285 appends a further starred copy for the unbounded case -- the same language with one more set of
  // positions; the compact form keeps the automaton (one cost byte per node and pending range on the GPU) smaller.
  Frag atom(const QAtom& a) {
    Frag f = empty();
    const bool unbounded = a.rmax == kUnbounded;
    const int plain = unbounded && a.rmin > 0 ? a.rmin - 1 : a.rmin;
    for (int k = 0; k < plain && !nfa_->too_large; k++) link(f, base(a));
    if (unbounded) {
      const Frag g = base(a);
      const int s = nfa_->add(), e = nfa_->add();
      nfa_->eps[size_t(s)].push_back(g.in);
      nfa_->eps[size_t(g.out)].push_back(e);
      if (a.rmin == 0) nfa_->eps[size_t(s)].push_back(e);       // zero times
      nfa_->eps[size_t(g.out)].push_back(g.in);                  // again
      link(f, Frag{s, e});
    } else {
      for (int k = a.rmin; k < a.rmax && !nfa_->too_large; k++) {     // max < min: no optional copies (compile_regexp.c:268)
        const Frag g = base(a);
        const int s = nfa_->add(), e = nfa_->add();
        nfa_->eps[size_t(s)].push_back(g.in);
        nfa_->eps[size_t(s)].push_back(e);
        nfa_->eps[size_t(g.out)].push_back(e);
        link(f, Frag{s, e});
      }
    }
    return f;
  }
  Frag sequence(const QSequence& s) {
    Frag f = empty();
    for (const QAtom& a : s.atoms) {
      if (nfa_->too_large) break;
      link(f, atom(a));
    }
    return f;
  }
  Frag regexp(const QRegexp& r) {
    if (r.choices.size() == 1) return sequence(r.choices[0]);
    const int s = nfa_->add(), e = nfa_->add();
    for (const QSequence& c : r.choices) {
      if (nfa_->too_large) break;
      const Frag g = sequence(c);
      nfa_->eps[size_t(s)].push_back(g.in);
      nfa_->eps[size_t(g.out)].push_back(e);
    }
    return {s, e};
  }
};

// pattern text -> tree; false with *err set on a syntax error
inline bool parse_query(const uint8_t* p, int64_t n, QRegexp* out, std::string* err) {
  if (n > kRegexMaxLen) { *err = "pattern text too long"; return false; }
  std::vector<QTok> toks;
  QueryLexer lx(p, n);
  if (!lx.run(&toks, err)) return false;
  QueryParser ps(toks);
  return ps.parse(out, err);
}

}  // namespace femto_amd
