// io.cpp -- the callers and data formats either side of the node-solve path:
//   mvx_read_lp / mvx_read_mps   replace glp_read_lp / glp_read_mps   (/root/reference/util.cpp:284,290)
//   mvx_bnb_write_events         the B&B event stream of message.cpp:32-191 (line format
//                                message.h:141-226) to a file / stdout sink instead of ZeroMQ
//   mvx_bnb_print_tree           the tree report of bs.cpp:329-343 / tree_print.h:12-22
//   mvx_bnb_solution_string      the solution line of bs.cpp:176-191
// The file formats are GLPK's (CPLEX LP and MPS); parsers here are written from the public format
// descriptions and cover the subset an MVOLPS model uses (objective, <=/>=/= rows, bounds,
// general / binary sections, MPS markers, RHS, RANGES, BOUNDS).
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/mvx_bnb.h"

namespace {

const double INF = HUGE_VAL;

std::string lower(std::string s) {
  for (auto &ch : s) ch = (char)std::tolower((unsigned char)ch);
  return s;
}

struct Model {
  int dir = MVX_MIN;
  double c0 = 0.0;
  std::vector<std::string> cols;
  std::map<std::string, int> col_id; // 1-based
  std::vector<double> obj;           // 1-based
  struct Row {
    std::string name;
    std::map<int, double> coef;
    int type = MVX_FR;
    double lb = 0, ub = 0;
  };
  std::vector<Row> rows;
  std::vector<int> ctype;
  std::vector<double> clb, cub;
  std::vector<int> kind;
  int col(const std::string &name) {
    auto it = col_id.find(name);
    if (it != col_id.end()) return it->second;
    cols.push_back(name);
    const int id = (int)cols.size();
    col_id[name] = id;
    obj.resize((size_t)id + 1, 0.0);
    ctype.resize((size_t)id + 1, MVX_LO);
    clb.resize((size_t)id + 1, 0.0);
    cub.resize((size_t)id + 1, INF);
    kind.resize((size_t)id + 1, MVX_CV);
    return id;
  }
};

void set_col_bounds(Model &M, int j, double lb, double ub) {
  M.clb[(size_t)j] = lb;
  M.cub[(size_t)j] = ub;
  if (lb == -INF && ub == INF) M.ctype[(size_t)j] = MVX_FR;
  else if (ub == INF) M.ctype[(size_t)j] = MVX_LO;
  else if (lb == -INF) M.ctype[(size_t)j] = MVX_UP;
  else if (lb == ub) M.ctype[(size_t)j] = MVX_FX;
  else M.ctype[(size_t)j] = MVX_DB;
}

int commit(const Model &M, mvx_prob *P) {
  const int n = (int)M.cols.size(), m = (int)M.rows.size();
  if (n < 1) return 1;
  mvx_erase_prob(P);
  mvx_set_obj_dir(P, M.dir);
  mvx_add_cols(P, n);
  if (m > 0) mvx_add_rows(P, m);
  mvx_set_obj_coef(P, 0, M.c0);
  for (int j = 1; j <= n; j++) {
    mvx_set_obj_coef(P, j, M.obj[(size_t)j]);
    mvx_set_col_name(P, j, M.cols[(size_t)j - 1].c_str());
    mvx_set_col_bnds(P, j, M.ctype[(size_t)j], M.clb[(size_t)j] == -INF ? 0.0 : M.clb[(size_t)j],
                     M.cub[(size_t)j] == INF ? 0.0 : M.cub[(size_t)j]);
    if (M.kind[(size_t)j] == MVX_IV) mvx_set_col_kind(P, j, MVX_IV);
  }
  std::vector<int> ind;
  std::vector<double> val;
  for (int i = 1; i <= m; i++) {
    const Model::Row &r = M.rows[(size_t)i - 1];
    ind.assign(1, 0);
    val.assign(1, 0.0);
    for (auto &kv : r.coef) {
      if (kv.second != 0.0) {
        ind.push_back(kv.first);
        val.push_back(kv.second);
      }
    }
    mvx_set_mat_row(P, i, (int)ind.size() - 1, ind.data(), val.data());
    mvx_set_row_bnds(P, i, r.type, r.lb, r.ub);
  }
  return 0;
}

// ------------------------------------------------------------------------- CPLEX LP
struct Tok {
  enum Kind { END, NUM, NAME, OP, COLON, PLUS, MINUS } kind = END;
  std::string text;
  double num = 0.0;
};

struct LpLexer {
  std::string s;
  size_t i = 0;
  explicit LpLexer(std::string text) : s(std::move(text)) {}
  static bool name_start(char ch) { return std::isalpha((unsigned char)ch) || std::strchr("_!\"#$%&(),;?@`'{}|~", ch) != nullptr; }
  static bool name_char(char ch) { return std::isalnum((unsigned char)ch) || std::strchr("_!\"#$%&(),.;?@`'{}|~[]", ch) != nullptr; }
  Tok next() {
    for (;;) {
      while (i < s.size() && std::isspace((unsigned char)s[i])) i++;
      if (i < s.size() && s[i] == '\\') { // comment to end of line
        while (i < s.size() && s[i] != '\n') i++;
        continue;
      }
      break;
    }
    Tok t;
    if (i >= s.size()) return t;
    const char ch = s[i];
    if (std::isdigit((unsigned char)ch) || (ch == '.' && i + 1 < s.size() && std::isdigit((unsigned char)s[i + 1]))) {
      char *end = nullptr;
      t.num = std::strtod(s.c_str() + i, &end);
      t.kind = Tok::NUM;
      i = (size_t)(end - s.c_str());
      return t;
    }
    if (ch == '+') { i++; t.kind = Tok::PLUS; return t; }
    if (ch == '-') { i++; t.kind = Tok::MINUS; return t; }
    if (ch == ':') { i++; t.kind = Tok::COLON; return t; }
    if (ch == '<' || ch == '>' || ch == '=') {
      t.kind = Tok::OP;
      t.text = std::string(1, ch);
      i++;
      if (i < s.size() && (s[i] == '=' || s[i] == '<' || s[i] == '>')) t.text += s[i++];
      if (t.text == "=<" || t.text == "<") t.text = "<=";
      if (t.text == "=>" || t.text == ">") t.text = ">=";
      if (t.text == "==") t.text = "=";
      return t;
    }
    if (name_start(ch)) {
      size_t j = i;
      while (j < s.size() && name_char(s[j])) j++;
      t.kind = Tok::NAME;
      t.text = s.substr(i, j - i);
      i = j;
      return t;
    }
    t.kind = Tok::NAME; // unknown character: surface it as a name so the parser reports it
    t.text = std::string(1, ch);
    i++;
    return t;
  }
};

enum Section { S_NONE, S_OBJ, S_ROWS, S_BOUNDS, S_GENERAL, S_BINARY, S_END };

int parse_lp(const std::string &text, Model &M, std::string &err) {
  LpLexer lx(text);
  std::vector<Tok> toks;
  for (Tok t = lx.next(); t.kind != Tok::END; t = lx.next()) toks.push_back(t);
  size_t k = 0;
  auto peek = [&](size_t d = 0) -> const Tok & {
    static const Tok endt;
    return k + d < toks.size() ? toks[k + d] : endt;
  };
  // section keyword at position k?  returns tokens consumed
  auto section_at = [&](Section &sec) -> int {
    if (peek().kind != Tok::NAME) return 0;
    const std::string w = lower(peek().text);
    const std::string w2 = peek(1).kind == Tok::NAME ? lower(peek(1).text) : "";
    if (w == "maximize" || w == "maximise" || w == "maximum" || w == "max") { sec = S_OBJ; M.dir = MVX_MAX; return 1; }
    if (w == "minimize" || w == "minimise" || w == "minimum" || w == "min") { sec = S_OBJ; M.dir = MVX_MIN; return 1; }
    if ((w == "subject" || w == "such") && (w2 == "to" || w2 == "that")) { sec = S_ROWS; return 2; }
    if (w == "st" || w == "s.t." || w == "st.") { sec = S_ROWS; return 1; }
    if (w == "bounds" || w == "bound") { sec = S_BOUNDS; return 1; }
    if (w == "general" || w == "generals" || w == "gen" || w == "integer" || w == "integers" || w == "int") { sec = S_GENERAL; return 1; }
    if (w == "binary" || w == "binaries" || w == "bin") { sec = S_BINARY; return 1; }
    if (w == "end") { sec = S_END; return 1; }
    return 0;
  };
  auto is_inf = [&](const Tok &t) { return t.kind == Tok::NAME && (lower(t.text) == "inf" || lower(t.text) == "infinity"); };
  // linear expression: returns false on syntax error; stops before an OP / section keyword / "name :"
  auto parse_expr = [&](std::map<int, double> &coef, double &constant) -> bool {
    bool any = false;
    for (;;) {
      Section tmp;
      if (peek().kind == Tok::END || peek().kind == Tok::OP) break;
      if (peek().kind == Tok::NAME && section_at(tmp)) break;
      if (peek().kind == Tok::NAME && peek(1).kind == Tok::COLON && any) break; // next row's label
      double sign = 1.0;
      bool signed_ = false;
      while (peek().kind == Tok::PLUS || peek().kind == Tok::MINUS) {
        if (peek().kind == Tok::MINUS) sign = -sign;
        signed_ = true;
        k++;
      }
      double num = 1.0;
      bool has_num = false;
      if (peek().kind == Tok::NUM) {
        num = peek().num;
        has_num = true;
        k++;
      }
      if (peek().kind == Tok::NAME && !is_inf(peek())) {
        Section t2;
        if (section_at(t2) || (peek(1).kind == Tok::COLON)) {
          if (!has_num) { if (signed_) return false; break; }
          constant += sign * num;
          any = true;
          break;
        }
        coef[M.col(peek().text)] += sign * num;
        k++;
      } else if (has_num) {
        constant += sign * num;
      } else {
        return false;
      }
      any = true;
    }
    return true;
  };

  Section sec = S_NONE;
  int autoname = 0;
  while (k < toks.size()) {
    Section s2 = sec;
    const int used = section_at(s2);
    if (used) {
      k += (size_t)used;
      sec = s2;
      if (sec == S_END) break;
      continue;
    }
    if (sec == S_OBJ) {
      if (peek().kind == Tok::NAME && peek(1).kind == Tok::COLON) k += 2; // objective name
      std::map<int, double> coef;
      double cst = 0.0;
      if (!parse_expr(coef, cst)) { err = "objective: syntax error"; return 1; }
      for (auto &kv : coef) M.obj[(size_t)kv.first] += kv.second;
      M.c0 += cst;
      if (peek().kind == Tok::OP) { err = "objective: unexpected relational operator"; return 1; }
      if (!section_at(s2) && peek().kind != Tok::END) { err = "objective: unexpected token '" + peek().text + "'"; return 1; }
    } else if (sec == S_ROWS) {
      Model::Row r;
      if (peek().kind == Tok::NAME && peek(1).kind == Tok::COLON) {
        r.name = peek().text;
        k += 2;
      } else
        r.name = "r." + std::to_string(++autoname);
      double cst = 0.0;
      if (!parse_expr(r.coef, cst)) { err = "constraint " + r.name + ": syntax error"; return 1; }
      if (peek().kind != Tok::OP) { err = "constraint " + r.name + ": missing relational operator"; return 1; }
      const std::string op = peek().text;
      k++;
      double sign = 1.0;
      while (peek().kind == Tok::PLUS || peek().kind == Tok::MINUS) {
        if (peek().kind == Tok::MINUS) sign = -sign;
        k++;
      }
      double rhs;
      if (peek().kind == Tok::NUM) rhs = sign * peek().num;
      else if (is_inf(peek())) rhs = sign * INF;
      else { err = "constraint " + r.name + ": missing right-hand side"; return 1; }
      k++;
      rhs -= cst;
      if (op == "<=") { r.type = MVX_UP; r.ub = rhs; }
      else if (op == ">=") { r.type = MVX_LO; r.lb = rhs; }
      else { r.type = MVX_FX; r.lb = r.ub = rhs; }
      M.rows.push_back(std::move(r));
    } else if (sec == S_BOUNDS) {
      // forms: [lo <=] x [<= up] | x >= lo | x = v | x free
      auto read_num = [&](double &v) -> bool {
        double sign = 1.0;
        size_t save = k;
        while (peek().kind == Tok::PLUS || peek().kind == Tok::MINUS) {
          if (peek().kind == Tok::MINUS) sign = -sign;
          k++;
        }
        if (peek().kind == Tok::NUM) { v = sign * peek().num; k++; return true; }
        if (is_inf(peek())) { v = sign * INF; k++; return true; }
        k = save;
        return false;
      };
      double lo = 0, v = 0;
      bool has_lo = read_num(lo);
      if (has_lo) {
        if (peek().kind != Tok::OP || peek().text == "=") { err = "bounds: expected '<=' or '>=' after number"; return 1; }
        const bool le = peek().text == "<=";
        k++;
        if (peek().kind != Tok::NAME) { err = "bounds: missing variable name"; return 1; }
        const int j = M.col(peek().text);
        k++;
        double lb = M.clb[(size_t)j], ub = M.cub[(size_t)j];
        if (le) lb = lo; else ub = lo;
        if (peek().kind == Tok::OP && peek().text != "=") {
          const bool le2 = peek().text == "<=";
          k++;
          if (!read_num(v)) { err = "bounds: missing upper bound"; return 1; }
          if (le2) ub = v; else lb = v;
        }
        set_col_bounds(M, j, lb, ub);
      } else {
        if (peek().kind != Tok::NAME) { err = "bounds: unexpected token"; return 1; }
        const int j = M.col(peek().text);
        k++;
        if (peek().kind == Tok::NAME && lower(peek().text) == "free") {
          k++;
          set_col_bounds(M, j, -INF, INF);
        } else if (peek().kind == Tok::OP) {
          const std::string op = peek().text;
          k++;
          if (!read_num(v)) { err = "bounds: missing bound value"; return 1; }
          double lb = M.clb[(size_t)j], ub = M.cub[(size_t)j];
          if (op == "<=") { ub = v; if (v < 0 && lb == 0.0) lb = -INF; }
          else if (op == ">=") lb = v;
          else lb = ub = v;
          set_col_bounds(M, j, lb, ub);
        } else { err = "bounds: expected operator after '" + M.cols[(size_t)j - 1] + "'"; return 1; }
      }
    } else if (sec == S_GENERAL || sec == S_BINARY) {
      if (peek().kind != Tok::NAME) { err = "integer section: unexpected token"; return 1; }
      const int j = M.col(peek().text);
      k++;
      M.kind[(size_t)j] = MVX_IV;
      if (sec == S_BINARY) set_col_bounds(M, j, 0.0, 1.0);
    } else {
      err = "missing 'maximize' or 'minimize' keyword";
      return 1;
    }
  }
  if (sec != S_END) { err = "missing 'end' keyword"; return 1; }
  return 0;
}

// ------------------------------------------------------------------------------ MPS
int parse_mps(std::istream &in, Model &M, std::string &err) {
  std::string line, section;
  std::map<std::string, int> row_id; // 1-based constraint rows; 0 = objective
  std::string obj_name;
  bool have_obj = false, integer_mode = false;
  std::map<std::string, char> row_kind;
  std::vector<bool> bounded; // column had an explicit bound
  while (std::getline(in, line)) {
    if (line.empty() || line[0] == '*') continue;
    std::istringstream ss(line);
    std::vector<std::string> f;
    for (std::string w; ss >> w;) f.push_back(w);
    if (f.empty()) continue;
    if (!std::isspace((unsigned char)line[0])) { // section header
      section = f[0];
      for (auto &ch : section) ch = (char)std::toupper((unsigned char)ch);
      if (section == "ENDATA") break;
      if (section == "OBJSENSE" && f.size() > 1) M.dir = (lower(f[1]).rfind("max", 0) == 0) ? MVX_MAX : MVX_MIN;
      continue;
    }
    if (section == "OBJSENSE") {
      M.dir = (lower(f[0]).rfind("max", 0) == 0) ? MVX_MAX : MVX_MIN;
    } else if (section == "ROWS") {
      if (f.size() < 2) { err = "ROWS: malformed line"; return 1; }
      const char t = (char)std::toupper((unsigned char)f[0][0]);
      row_kind[f[1]] = t;
      if (t == 'N') {
        if (!have_obj) { have_obj = true; obj_name = f[1]; row_id[f[1]] = 0; }
        else row_id[f[1]] = -1; // extra free rows are dropped
      } else {
        Model::Row r;
        r.name = f[1];
        r.type = t == 'L' ? MVX_UP : t == 'G' ? MVX_LO : MVX_FX;
        M.rows.push_back(r);
        row_id[f[1]] = (int)M.rows.size();
      }
    } else if (section == "COLUMNS") {
      if (f.size() >= 3 && f[1] == "'MARKER'") {
        integer_mode = (f[2] == "'INTORG'");
        continue;
      }
      if (f.size() < 3 || (f.size() % 2) == 0) { err = "COLUMNS: malformed line"; return 1; }
      const int j = M.col(f[0]);
      if (integer_mode) M.kind[(size_t)j] = MVX_IV;
      for (size_t t = 1; t + 1 < f.size(); t += 2) {
        auto it = row_id.find(f[t]);
        if (it == row_id.end()) { err = "COLUMNS: unknown row " + f[t]; return 1; }
        const double v = std::strtod(f[t + 1].c_str(), nullptr);
        if (it->second == 0) M.obj[(size_t)j] += v;
        else if (it->second > 0) M.rows[(size_t)it->second - 1].coef[j] += v;
      }
    } else if (section == "RHS" || section == "RANGES") {
      size_t t0 = (f.size() % 2) ? 1 : 0; // optional set name
      for (size_t t = t0; t + 1 < f.size(); t += 2) {
        auto it = row_id.find(f[t]);
        if (it == row_id.end()) { err = section + ": unknown row " + f[t]; return 1; }
        const double v = std::strtod(f[t + 1].c_str(), nullptr);
        if (it->second == 0) { if (section == "RHS") M.c0 = -v; continue; }
        if (it->second < 0) continue;
        Model::Row &r = M.rows[(size_t)it->second - 1];
        if (section == "RHS") {
          if (r.type == MVX_UP) r.ub = v;
          else if (r.type == MVX_LO) r.lb = v;
          else r.lb = r.ub = v;
        } else {
          const char kd = row_kind[f[t]];
          if (kd == 'L') { r.lb = r.ub - std::fabs(v); r.type = MVX_DB; }
          else if (kd == 'G') { r.ub = r.lb + std::fabs(v); r.type = MVX_DB; }
          else if (kd == 'E') { if (v >= 0) r.ub = r.lb + v; else r.lb = r.ub + v; r.type = (r.lb == r.ub) ? MVX_FX : MVX_DB; }
        }
      }
    } else if (section == "BOUNDS") {
      if (f.size() < 3) { err = "BOUNDS: malformed line"; return 1; }
      std::string bt = f[0];
      for (auto &ch : bt) ch = (char)std::toupper((unsigned char)ch);
      // "type set column [value]" or "type column [value]" when the set name is omitted
      size_t ci = (f.size() >= 4 || bt == "FR" || bt == "MI" || bt == "PL" || bt == "BV") && M.col_id.count(f[2]) ? 2 : 1;
      if (!M.col_id.count(f[ci])) { err = "BOUNDS: unknown column " + f[ci]; return 1; }
      const int j = M.col_id[f[ci]];
      const double v = ci + 1 < f.size() ? std::strtod(f[ci + 1].c_str(), nullptr) : 0.0;
      double lb = M.clb[(size_t)j], ub = M.cub[(size_t)j];
      if (bt == "UP") { ub = v; if (v < 0 && lb == 0.0) lb = -INF; }
      else if (bt == "LO") lb = v;
      else if (bt == "FX") lb = ub = v;
      else if (bt == "FR") { lb = -INF; ub = INF; }
      else if (bt == "MI") lb = -INF;
      else if (bt == "PL") ub = INF;
      else if (bt == "BV") { lb = 0; ub = 1; M.kind[(size_t)j] = MVX_IV; }
      else if (bt == "LI") { lb = v; M.kind[(size_t)j] = MVX_IV; }
      else if (bt == "UI") { ub = v; M.kind[(size_t)j] = MVX_IV; }
      else { err = "BOUNDS: unknown bound type " + bt; return 1; }
      set_col_bounds(M, j, lb, ub);
    }
  }
  if (!have_obj) { err = "no objective (N) row"; return 1; }
  for (auto &r : M.rows) { // rows without an RHS entry default to 0
    if (r.type == MVX_UP) r.lb = 0;
    if (r.type == MVX_LO) r.ub = 0;
  }
  return 0;
}

std::string fmt(double v) { // ostream default formatting, as sstr()/operator<< give in the reference
  std::ostringstream o;
  o << v;
  return o.str();
}

} // namespace

extern "C" {

int mvx_read_lp(mvx_prob *P, const void * /*parm*/, const char *fname) {
  std::ifstream f(fname);
  if (!f) {
    std::fprintf(stderr, "mvx_read_lp: cannot open '%s'\n", fname);
    return 1;
  }
  std::stringstream ss;
  ss << f.rdbuf();
  Model M;
  std::string err;
  if (parse_lp(ss.str(), M, err) || commit(M, P)) {
    std::fprintf(stderr, "mvx_read_lp: %s: %s\n", fname, err.empty() ? "empty model" : err.c_str());
    return 1;
  }
  return 0;
}

int mvx_read_mps(mvx_prob *P, int /*fmt*/, const void * /*parm*/, const char *fname) {
  std::ifstream f(fname);
  if (!f) {
    std::fprintf(stderr, "mvx_read_mps: cannot open '%s'\n", fname);
    return 1;
  }
  Model M;
  std::string err;
  if (parse_mps(f, M, err) || commit(M, P)) {
    std::fprintf(stderr, "mvx_read_mps: %s: %s\n", fname, err.empty() ? "empty model" : err.c_str());
    return 1;
  }
  return 0;
}

// one text line per event: "timeSpan nodeType oid pid direction [LPBound [sumInfeas nViolated] [bCond eCond]]"
// (message.h:116-139,141-226; field sets per event type message.cpp:49-162; bCond/eCond are the
// constants 1 and 2 the reference sends, bs.cpp:126-127,201-202,242-243).  Ends with "END"
// (message.h:230-236).
int mvx_bnb_write_events(const mvx_bnb_result *res, const char *path) {
  static const char *names[] = {"pregnant", "integer", "infeasible", "fathomed", "branched", "candidate"};
  static const char *dirs[] = {"M", "R", "L"};
  FILE *f = path ? std::fopen(path, "w") : stdout;
  if (!f) return 1;
  for (int k = 0; k < res->n_events; k++) {
    const mvx_bnb_event &e = res->events[k];
    std::string line = fmt((double)k) + " " + names[e.type] + " " + std::to_string(e.oid) + " " + std::to_string(e.pid) + " " + dirs[e.direction];
    switch (e.type) {
      case MVX_EV_PREGNANT: line += " " + fmt(e.lp_bound) + " 1 2"; break;
      case MVX_EV_INTEGER: line += " " + fmt(e.lp_bound); break;
      case MVX_EV_INFEASIBLE: line += " 1 2"; break;
      case MVX_EV_FATHOMED: break;
      case MVX_EV_BRANCHED: line += " " + fmt(e.lp_bound) + " " + fmt(e.sum_infeas) + " " + std::to_string(e.n_violated) + " 1 2"; break;
      case MVX_EV_CANDIDATE: line += " " + fmt(e.lp_bound); break;
    }
    std::fprintf(f, "%s\n", line.c_str());
  }
  std::fprintf(f, "END\n");
  if (path) std::fclose(f);
  return 0;
}

// bs.cpp:329-343 + tree_print.h:12-22: pre-order walk, one space per depth level, "-<oid>[ I|F|B]"
int mvx_bnb_print_tree(const mvx_bnb_result *res, const char *path) {
  FILE *f = path ? std::fopen(path, "w") : stdout;
  if (!f) return 1;
  const int nn = res->n_nodes;
  std::vector<std::vector<int>> kids((size_t)nn + 1);
  for (int oid = 2; oid <= nn; oid++) kids[(size_t)res->parent[oid]].push_back(oid);
  std::fprintf(f, "[I = Integral node, F = Infeasible node, B = Worse bound node]\n");
  std::vector<std::pair<int, int>> stack;
  if (nn >= 1) stack.push_back({1, 0});
  while (!stack.empty()) {
    const auto [oid, depth] = stack.back();
    stack.pop_back();
    for (int i = 0; i < depth; i++) std::fputc(' ', f);
    const int pr = res->prune[oid];
    std::fprintf(f, "-%d%s\n", oid, pr == 0 ? " I" : pr == 1 ? " F" : pr == 3 ? " B" : "");
    for (auto it = kids[(size_t)oid].rbegin(); it != kids[(size_t)oid].rend(); ++it) stack.push_back({*it, depth + 1});
  }
  if (path) std::fclose(f);
  return 0;
}

// bs.cpp:176-191: "[oid] Solution is: c*(x[i] = v) + ... c0 = obj"
int mvx_bnb_solution_string(const mvx_lp_api *api, const void *root, const mvx_bnb_result *res, char *buf, int cap) {
  if (!api) api = mvx_hip_lp_api();
  std::string s;
  if (res->has_incumbent) {
    s = "[" + std::to_string(res->incumbent_oid) + "] Solution is: ";
    for (int i = 1; i <= res->n; i++) {
      const double c = api->get_obj_coef(root, i);
      if (res->x[i] != 0 && c != 0) s += fmt(c) + "*(x[" + std::to_string(i) + "] = " + fmt(res->x[i]) + ") + ";
    }
    s += fmt(api->get_obj_coef(root, 0)) + " = " + fmt(res->best_lower) + "\n";
  }
  if ((int)s.size() + 1 > cap) return (int)s.size() + 1;
  std::memcpy(buf, s.c_str(), s.size() + 1);
  return 0;
}

} // extern "C"
