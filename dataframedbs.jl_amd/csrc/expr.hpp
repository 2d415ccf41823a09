// expr.hpp — the engine's own reading of the expression IR (include/dfdb_ir.h): typed tree with Julia
// result-type inference (the role Base._return_type plays in src/tables/broadcast.jl:13) and the pattern
// matching that routes a predicate to the specialised scan kernels or to the device interpreter.
#pragma once
#include "common.hpp"
#include "kernels.hpp"

struct dfdb_table;

namespace dfdb {

struct Node {
  int op = 0;
  int32_t dtype = 0;             // inferred result dtype
  int col = -1;                  // DFIR_COL
  uint64_t cbits = 0;            // DFIR_CONST value bit pattern (dtype = const dtype)
  std::string str;               // DFIR_CONST_STR
  std::vector<uint64_t> set;     // DFIR_CONST_SET values (bit patterns of set_dtype)
  int32_t set_dtype = 0;
  int cast_to = 0;
  std::unique_ptr<Node> a, b;
  std::unique_ptr<Node> clone() const;
};
using NodePtr = std::unique_ptr<Node>;

NodePtr parse_ir(const dfdb_table& t, const uint8_t* ir, size_t len);
NodePtr make_and(NodePtr a, NodePtr b);            // BlockBroadcasting(&, (old, new)): selection.jl:44-47
void required_columns(const Node& n, std::vector<int>& out);  // first-appearance order, unique

int promote_num(int a, int b);

// flatten a tree of DFIR_AND into its conjuncts (left to right)
void flatten_and(const Node& n, std::vector<const Node*>& out);

// `col OP const` / `const OP col` over a numeric column whose constant converts exactly to the column's
// own type; fills term (col pointer left null: the caller resolves the ordinal) and returns the ordinal.
// always: set to +1/-1 when the comparison is constant true/false for every value of the column type.
bool match_simple_term(const Node& n, const dfdb_table& t, ScanTerm& term, int& ordinal);
bool match_column_transform(const Node* e, ScanTerm& out, const Node*& coln);   // the column itself / rem / col * k + d / col / k (ScanTerm::pre)

// `strcol == "x"`, `!=`, startswith, endswith  -> mode 0..3 (launch_str_match)
bool match_string_term(const Node& n, const dfdb_table& t, int& ordinal, int& mode, std::string& pat, bool allow_nullable = false);

}  // namespace dfdb
