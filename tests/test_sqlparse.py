"""CPU: the SQL front-end mirrors sqlparser's expression grammar (precedence, associativity, Nested, literals)."""
import pytest

from chapterhouseqe_amd import sqlast as A
from chapterhouseqe_amd.sqlparse import SqlParseError, parse_expr, parse_select, parse_statements

B = A.BinaryOperator


def test_precedence_and_or():
    e = parse_expr("a + b > c and d < 5.0 or e > 1.0")
    assert isinstance(e, A.BinaryOp) and e.op == B.Or
    assert e.left.op == B.And and e.left.left.op == B.Gt and e.left.left.left.op == B.Plus
    assert e.right.op == B.Gt


def test_mul_binds_tighter_and_left_assoc():
    e = parse_expr("a+1.0/(2.0+c)*b")       # record_utils/test_compute_value.rs:131-133
    assert e.op == B.Plus and e.right.op == B.Multiply and e.right.left.op == B.Divide
    assert isinstance(e.right.left.right, A.Nested)
    e = parse_expr("a / b / c")
    assert e.op == B.Divide and e.left.op == B.Divide


def test_literals_keep_text():
    assert parse_expr("10.0") == A.number("10.0")
    assert parse_expr("1.") == A.number("1.")
    assert parse_expr(".5") == A.number(".5")
    assert parse_expr("7L") == A.number("7", True)
    assert parse_expr("'it''s'") == A.string("it's")
    assert parse_expr("TRUE") == A.boolean(True)
    assert isinstance(parse_expr("null").value, A.UnsupportedValue)


def test_identifiers():
    assert parse_expr("table_b.text") == A.compound("table_b", "text")
    assert parse_expr('"My Col"') == A.Identifier(A.Ident("My Col", '"'))


def test_unsupported_nodes_are_kept():
    assert isinstance(parse_expr("-5"), A.UnsupportedExpr)
    assert isinstance(parse_expr("not a"), A.UnsupportedExpr)
    assert isinstance(parse_expr("abs(a)"), A.UnsupportedExpr)
    assert parse_expr("a - 1").op == B.Minus
    assert parse_expr("a || b").op == B.StringConcat


def test_select_items_and_aliases():
    s = parse_select("select id, value1 v, id + 10.0 as id_plus_10, t.*, * from read_files('x/*.parquet') tbl where id % 2 = 0;")
    kinds = [type(i).__name__ for i in s.projection]
    assert kinds == ["UnnamedExpr", "ExprWithAlias", "ExprWithAlias", "QualifiedWildcard", "Wildcard"]
    assert s.from_.func_name == "read_files" and s.from_.args == ("x/*.parquet",) and s.from_.alias == "tbl"
    assert s.selection.op == B.Eq


def test_statement_split_and_comments():
    stmts = parse_statements("-- query 1\nselect * from read_files('a') where id < 25;\n\n-- q2\nselect id from read_files('b');")
    assert len(stmts) == 2 and stmts[1].selection is None


def test_errors():
    with pytest.raises(SqlParseError):
        parse_expr("a +")
    with pytest.raises(SqlParseError):
        parse_expr("(a + b")
