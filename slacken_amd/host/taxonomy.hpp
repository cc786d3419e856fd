// taxonomy.hpp -- host-side mirror of com.jnpersson.slacken.Taxonomy and KrakenReport (C++17, header only).
// Reference: S/slacken/Taxonomy.scala:29-137,159-244 and S/slacken/KrakenReport.scala:26-116
// (S/ = src/main/scala/com/jnpersson/ under /root/reference).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace slk_host {

using Taxon = int32_t;
constexpr Taxon NONE = 0, ROOT = 1;  // Taxonomy.scala:30-31

// Taxonomy.Rank (Taxonomy.scala:33-48): index = depth + 1 ("unclassified" has depth -1); NO_RANK for any other title
constexpr int NO_RANK = -2;
inline const char *rank_titles(int i) {
  static const char *t[] = {"unclassified", "root", "superkingdom", "kingdom", "phylum", "class", "order", "family", "genus", "species"};
  return t[i];
}
inline const char *rank_codes(int i) {
  static const char *c[] = {"U", "R", "D", "K", "P", "C", "O", "F", "G", "S"};
  return c[i];
}
inline int rank_index(const std::string &title) {  // Taxonomy.rank :55-67; returns depth + 1, or NO_RANK
  for (int i = 0; i < 10; i++) if (title == rank_titles(i)) return i;
  return NO_RANK;
}

inline std::string trim(const std::string &s) {
  size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
  return a == std::string::npos ? "" : s.substr(a, b - a + 1);
}
inline std::vector<std::string> split_pipe(const std::string &line) {  // _.split("\\|")
  std::vector<std::string> out;
  std::string cur;
  for (char ch : line) {
    if (ch == '|') { out.push_back(cur); cur.clear(); }
    else cur.push_back(ch);
  }
  out.push_back(cur);
  return out;
}

struct Taxonomy {
  std::vector<Taxon> parents;        // Taxonomy.parents :159
  std::vector<int8_t> ranks;         // rank index (depth + 1) or NO_RANK (null in the reference)
  std::vector<std::string> names;    // scientificNames (empty = null)
  std::vector<uint8_t> has_name;
  std::vector<Taxon> primary;
  mutable std::vector<std::vector<Taxon>> children_;

  Taxon size() const { return (Taxon)parents.size(); }
  bool isDefined(Taxon t) const { return parents[t] != NONE || t == ROOT; }  // :171-172

  // Taxonomy.fromNodesAndNames :81-109
  static Taxonomy fromNodesAndNames(const std::vector<std::tuple<Taxon, Taxon, std::string>> &nodes,
                                    const std::vector<std::pair<Taxon, std::string>> &names,
                                    const std::vector<std::pair<Taxon, Taxon>> &merged) {
    Taxon max1 = 0, max2 = 0;
    for (auto &n : nodes) max1 = std::max(max1, std::get<0>(n) + 1);
    for (auto &m : merged) max2 = std::max(max2, m.first + 1);
    Taxon num = std::max(std::max(max1, max2), (Taxon)2);
    Taxonomy t;
    t.names.assign(num, "");
    t.has_name.assign(num, 0);
    for (auto &n : names) if (n.first >= 0 && n.first < num) { t.names[n.first] = n.second; t.has_name[n.first] = 1; }
    t.names[NONE] = "unclassified";
    t.has_name[NONE] = 1;
    t.parents.assign(num, NONE);
    t.ranks.assign(num, (int8_t)NO_RANK);
    for (auto &n : nodes) {
      t.parents[std::get<0>(n)] = std::get<1>(n);
      t.ranks[std::get<0>(n)] = (int8_t)rank_index(std::get<2>(n));
    }
    t.primary.resize(num);
    for (Taxon i = 0; i < num; i++) t.primary[i] = i;
    for (auto &m : merged) t.primary[m.first] = m.second;
    t.parents[ROOT] = NONE;
    t.ranks[NONE] = 0;  // Unclassified
    t.ranks[ROOT] = 1;  // Root
    return t;
  }

  // Taxonomy.load :116-137 (nodes.dmp, names.dmp, optional merged.dmp)
  static Taxonomy load(const std::string &dir) {
    std::vector<std::tuple<Taxon, Taxon, std::string>> nodes;
    std::vector<std::pair<Taxon, std::string>> names;
    std::vector<std::pair<Taxon, Taxon>> merged;
    std::string line;
    std::ifstream fn(dir + "/nodes.dmp");
    if (!fn) throw std::runtime_error("cannot open " + dir + "/nodes.dmp");
    while (std::getline(fn, line)) {
      auto x = split_pipe(line);
      if (x.size() < 3) continue;
      nodes.emplace_back(std::stoi(trim(x[0])), std::stoi(trim(x[1])), trim(x[2]));
    }
    std::ifstream fa(dir + "/names.dmp");
    if (!fa) throw std::runtime_error("cannot open " + dir + "/names.dmp");
    while (std::getline(fa, line)) {
      auto x = split_pipe(line);
      if (x.size() < 4) continue;
      if (trim(x[3]) == "scientific name") names.emplace_back(std::stoi(trim(x[0])), trim(x[1]));
    }
    std::ifstream fm(dir + "/merged.dmp");
    while (fm && std::getline(fm, line)) {
      auto x = split_pipe(line);
      if (x.size() < 2) continue;
      merged.emplace_back(std::stoi(trim(x[0])), std::stoi(trim(x[1])));
    }
    return fromNodesAndNames(nodes, names, merged);
  }

  // Taxonomy.depth :217-224: the depth of the nearest ranked ancestor-or-self; NONE = -1
  int depth(Taxon t) const {
    while (t > 0 && t < size()) {
      if (ranks[t] != NO_RANK) return ranks[t] - 1;
      t = parents[t];
    }
    return -1;
  }

  // Taxonomy.taxaWithDescendants :304-311 as a membership vector
  std::vector<uint8_t> withDescendants(const std::vector<Taxon> &taxa) const {
    std::vector<uint8_t> in(parents.size(), 0);
    std::vector<Taxon> stack(taxa.begin(), taxa.end());
    while (!stack.empty()) {
      Taxon t = stack.back();
      stack.pop_back();
      if (t < 0 || t >= size() || in[t]) continue;
      in[t] = 1;
      for (Taxon c : children()[t]) stack.push_back(c);
    }
    return in;
  }

  // Taxonomy.children :186-195: built by PREPENDING while iterating taxids upward => each list is in descending id order
  // (here: appended while iterating downward -- the same lists without the quadratic cost for taxa with 1e5 children)
  const std::vector<std::vector<Taxon>> &children() const {
    if (children_.empty()) {
      children_.assign(parents.size(), {});
      for (Taxon t = size() - 1; t >= 0; t--)
        if (isDefined(t)) children_[parents[t]].push_back(t);
    }
    return children_;
  }
};

// Java's "%6.2f".format(x): java.util.Formatter takes the SHORTEST decimal digits that identify the double (the digits of
// Double.toString) and rounds those HALF_UP at the requested precision (FormattedFloatingDecimal.applyPrecision).  C's printf
// rounds the exact binary value instead, which differs both on exact ties (0.125 -> 0.13) and on decimal ties that are not
// binary ties (12.345 -> 12.35; 99.995 -> 100.00).  x >= 0 and finite.
inline std::string java_format_6_2f(double x) {
  char b[64];
  if (!(x >= 0) || x > 1e15) { snprintf(b, sizeof b, "%6.2f", x); return b; }
  if (x == 0) return "  0.00";
  int prec = 1;
  for (; prec <= 17; prec++) {
    snprintf(b, sizeof b, "%.*e", prec - 1, x);
    if (strtod(b, nullptr) == x) break;
  }
  std::string digits;               // d.ddddde[+-]XX  ->  digits, value = 0.digits * 10^decExp
  const char *p = b;
  for (; *p && *p != 'e'; p++) if (*p != '.') digits.push_back(*p);
  int decExp = atoi(p + 1) + 1;
  int keep = decExp + 2;            // number of leading digits kept (may be <= 0)
  unsigned __int128 cents = 0;
  if (keep >= (int)digits.size()) {
    for (char c : digits) cents = cents * 10 + (unsigned)(c - '0');
    for (int i = (int)digits.size(); i < keep; i++) cents *= 10;
  } else if (keep >= 0) {
    for (int i = 0; i < keep; i++) cents = cents * 10 + (unsigned)(digits[i] - '0');
    if (digits[keep] >= '5') cents += 1;  // HALF_UP on the decimal digits
  }
  snprintf(b, sizeof b, "%3llu.%02llu", (unsigned long long)(cents / 100), (unsigned long long)(cents % 100));
  return b;
}

// KrakenReport (KrakenReport.scala:44-116), non-compatible format (with the header line), reportZeros = false
struct KrakenReport {
  const Taxonomy &tax;
  std::map<Taxon, long> taxonCounts, cladeTotals;
  long totalSequences = 0;

  KrakenReport(const Taxonomy &t, const std::vector<std::pair<Taxon, long>> &counts) : tax(t) {
    for (auto &c : counts) {  // TreeAggregator :27-41
      taxonCounts[c.first] += c.second;
      totalSequences += c.second;
      for (Taxon p = c.first; p != NONE; p = (p >= 0 && p < tax.size()) ? tax.parents[p] : NONE) cladeTotals[p] += c.second;
      if (c.first == NONE) cladeTotals[NONE] = c.second;
    }
  }
  long clade(Taxon t) const { auto it = cladeTotals.find(t); return it == cladeTotals.end() ? 0 : it->second; }
  long own(Taxon t) const { auto it = taxonCounts.find(t); return it == taxonCounts.end() ? 0 : it->second; }

  std::string reportLine(Taxon taxid, int rank, int rankDepth, int depth) const {  // :72-77
    std::ostringstream o;
    o << java_format_6_2f(100.0 * (double)clade(taxid) / (double)totalSequences) << '\t' << clade(taxid) << '\t' << own(taxid)
      << '\t' << rank_codes(rank);
    if (rankDepth != 0) o << rankDepth;
    o << '\t' << taxid << '\t' << std::string(2 * depth, ' ')
      << ((taxid >= 0 && taxid < tax.size() && tax.has_name[taxid]) ? tax.names[taxid] : "");
    return o.str();
  }
  void dfs(std::ostream &out, Taxon taxid, int rank, int rankDepth, int depth) const {  // :82-102
    int r = rank, rd = rankDepth + 1;
    if (tax.ranks[taxid] != NO_RANK) { r = tax.ranks[taxid]; rd = 0; }
    out << reportLine(taxid, r, rd, depth) << '\n';
    std::vector<std::pair<Taxon, long>> kids;
    for (Taxon c : tax.children()[taxid]) kids.emplace_back(c, clade(c));
    std::stable_sort(kids.begin(), kids.end(), [](auto &a, auto &b) { return a.second > b.second; });  // sortWith(_._2 > _._2)
    for (auto &kc : kids) if (kc.second > 0) dfs(out, kc.first, r, rd, depth + 1);
  }
  void print(std::ostream &out) const {  // :104-116
    out << "#Perc\tAggregate\tIn taxon\tRank\tTaxon\tName\n";
    if (own(NONE) != 0) out << reportLine(NONE, 0, 0, 0) << '\n';
    dfs(out, ROOT, 1, 0, 0);
  }
};

}  // namespace slk_host
