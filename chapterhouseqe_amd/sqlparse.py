"""Minimal SQL front-end producing the `sqlast` nodes the record kernels consume.

The reference gets its expressions from the third-party `sqlparser 0.52` crate (GenericDialect,
reference: src/planner/logical_planner.rs:228-300; test use: record_utils/test_compute_value.rs:127-148).
That crate is control-plane and out of scope; this module only restates enough of its *expression*
grammar -- tokens, operator precedence (Or 5 < And 10 < comparison 20 < +,- 30 < *,/,% 40), left
associativity, `Nested` for parentheses, `Number` text kept verbatim -- for tests, the bench and the
sample queries (reference: sample_queries/simple.sql) to be written as SQL text.

    select <items> from <func>('<path>') [[as] alias] [where <expr>]
"""
from __future__ import annotations

import re
from dataclasses import dataclass
from typing import List, Optional, Tuple

from . import sqlast as A

_TOKEN_RE = re.compile(
    r"""\s*(?:
        (?P<comment>--[^\n]*)
      | (?P<number>(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?)(?P<long>L)?
      | (?P<string>'(?:[^']|'')*')
      | (?P<qident>"(?:[^"]|"")*")
      | (?P<word>[A-Za-z_][A-Za-z0-9_$]*)
      | (?P<op><>|!=|<=|>=|\|\||==|[-+*/%=<>(),.;&|^])
    )""",
    re.X,
)

_PREC_OR, _PREC_AND, _PREC_NOT, _PREC_EQ = 5, 10, 15, 20
_PREC_PIPE, _PREC_CARET, _PREC_AMP, _PREC_XOR = 21, 22, 23, 24
_PREC_PLUS, _PREC_MUL = 30, 40

_BINOPS = {
    "=": (A.BinaryOperator.Eq, _PREC_EQ), "==": (A.BinaryOperator.Eq, _PREC_EQ),
    "<>": (A.BinaryOperator.NotEq, _PREC_EQ), "!=": (A.BinaryOperator.NotEq, _PREC_EQ),
    "<": (A.BinaryOperator.Lt, _PREC_EQ), "<=": (A.BinaryOperator.LtEq, _PREC_EQ),
    ">": (A.BinaryOperator.Gt, _PREC_EQ), ">=": (A.BinaryOperator.GtEq, _PREC_EQ),
    "+": (A.BinaryOperator.Plus, _PREC_PLUS), "-": (A.BinaryOperator.Minus, _PREC_PLUS),
    "*": (A.BinaryOperator.Multiply, _PREC_MUL), "/": (A.BinaryOperator.Divide, _PREC_MUL),
    "%": (A.BinaryOperator.Modulo, _PREC_MUL), "||": (A.BinaryOperator.StringConcat, _PREC_MUL),
    "|": (A.BinaryOperator.BitwiseOr, _PREC_PIPE), "^": (A.BinaryOperator.BitwiseXor, _PREC_CARET),
    "&": (A.BinaryOperator.BitwiseAnd, _PREC_AMP),
}
_WORD_BINOPS = {"OR": (A.BinaryOperator.Or, _PREC_OR), "AND": (A.BinaryOperator.And, _PREC_AND),
                "XOR": (A.BinaryOperator.Xor, _PREC_XOR)}
_STOP_WORDS = {"FROM", "WHERE", "AS", "GROUP", "ORDER", "LIMIT"}


class SqlParseError(ValueError):
    pass


@dataclass(frozen=True)
class TableFunc:
    """`read_files('glob') alias` -- reference: planner OperatorTask::TableFunc { alias, func_name, args }."""
    func_name: str
    args: Tuple[str, ...]
    alias: Optional[str]


@dataclass(frozen=True)
class Select:
    projection: Tuple[A.SelectItem, ...]
    from_: Optional[TableFunc]
    selection: Optional[A.Expr]


def _tokenize(text: str) -> List[Tuple[str, str]]:
    out, pos = [], 0
    while pos < len(text):
        m = _TOKEN_RE.match(text, pos)
        if not m or m.end() == pos:
            if text[pos:].strip() == "":
                break
            raise SqlParseError(f"cannot tokenize at: {text[pos:pos + 20]!r}")
        pos = m.end()
        if m.group("comment") is not None:
            continue
        if m.group("number") is not None:
            out.append(("numberL" if m.group("long") else "number", m.group("number")))
        elif m.group("string") is not None:
            out.append(("string", m.group("string")[1:-1].replace("''", "'")))
        elif m.group("qident") is not None:
            out.append(("qident", m.group("qident")[1:-1].replace('""', '"')))
        elif m.group("word") is not None:
            out.append(("word", m.group("word")))
        else:
            out.append(("op", m.group("op")))
    return out


class _Parser:
    def __init__(self, toks):
        self.toks, self.i = toks, 0

    def peek(self):
        return self.toks[self.i] if self.i < len(self.toks) else ("eof", "")

    def next(self):
        t = self.peek()
        self.i += 1
        return t

    def accept_op(self, op):
        if self.peek() == ("op", op):
            self.i += 1
            return True
        return False

    def accept_word(self, w):
        k, v = self.peek()
        if k == "word" and v.upper() == w:
            self.i += 1
            return True
        return False

    def expect_op(self, op):
        if not self.accept_op(op):
            raise SqlParseError(f"expected {op!r}, found {self.peek()[1]!r}")

    # ---- expressions: sqlparser Parser::parse_subexpr ------------------------------------------
    def next_precedence(self):
        k, v = self.peek()
        if k == "op" and v in _BINOPS:
            return _BINOPS[v][1]
        if k == "word" and v.upper() in _WORD_BINOPS:
            return _WORD_BINOPS[v.upper()][1]
        return 0

    def parse_expr(self, precedence=0) -> A.Expr:
        expr = self.parse_prefix()
        while True:
            nxt = self.next_precedence()
            if precedence >= nxt:
                break
            k, v = self.next()
            op = _BINOPS[v][0] if k == "op" else _WORD_BINOPS[v.upper()][0]
            right = self.parse_expr(nxt)
            expr = A.BinaryOp(expr, op, right)
        return expr

    def parse_prefix(self) -> A.Expr:
        k, v = self.next()
        if k == "number":
            return A.ValueExpr(A.Number(v, False))
        if k == "numberL":
            return A.ValueExpr(A.Number(v, True))
        if k == "string":
            return A.ValueExpr(A.SingleQuotedString(v))
        if k == "op" and v == "(":
            inner = self.parse_expr(0)
            self.expect_op(")")
            return A.Nested(inner)
        if k == "op" and v in ("-", "+"):
            inner = self.parse_expr(_PREC_MUL)
            return A.UnsupportedExpr(f"UnaryOp {{ op: {'Minus' if v == '-' else 'Plus'}, expr: {inner!r} }}")
        if k == "word":
            up = v.upper()
            if up == "TRUE":
                return A.ValueExpr(A.Boolean(True))
            if up == "FALSE":
                return A.ValueExpr(A.Boolean(False))
            if up == "NULL":
                return A.ValueExpr(A.UnsupportedValue("Null"))
            if up == "NOT":
                inner = self.parse_expr(_PREC_NOT)
                return A.UnsupportedExpr(f"UnaryOp {{ op: Not, expr: {inner!r} }}")
            if self.peek() == ("op", "("):
                depth = 0
                while True:  # skip the call, it is rejected by compute_value anyway
                    kk, vv = self.next()
                    if kk == "eof":
                        raise SqlParseError("unterminated function call")
                    if (kk, vv) == ("op", "("):
                        depth += 1
                    if (kk, vv) == ("op", ")"):
                        depth -= 1
                        if depth == 0:
                            break
                return A.UnsupportedExpr(f"Function({v})")
            return self._identifier_tail(A.Ident(v))
        if k == "qident":
            return self._identifier_tail(A.Ident(v, '"'))
        raise SqlParseError(f"unexpected token {v!r}")

    def _identifier_tail(self, first: A.Ident) -> A.Expr:
        parts = [first]
        while self.peek() == ("op", "."):
            self.next()
            k, v = self.next()
            if k == "word":
                parts.append(A.Ident(v))
            elif k == "qident":
                parts.append(A.Ident(v, '"'))
            else:
                raise SqlParseError("expected identifier after '.'")
        if len(parts) == 1:
            return A.Identifier(parts[0])
        return A.CompoundIdentifier(tuple(parts))

    # ---- select ---------------------------------------------------------------------------------
    def parse_select_item(self) -> A.SelectItem:
        if self.accept_op("*"):
            return A.Wildcard()
        # qualified wildcard: ident . *
        if self.peek()[0] in ("word", "qident") and self.i + 2 < len(self.toks) + 1:
            j = self.i
            if (j + 2 < len(self.toks) and self.toks[j + 1] == ("op", ".") and self.toks[j + 2] == ("op", "*")):
                prefix = self.toks[j][1]
                self.i += 3
                return A.QualifiedWildcard(prefix)
        expr = self.parse_expr(0)
        if self.accept_word("AS"):
            k, v = self.next()
            if k not in ("word", "qident"):
                raise SqlParseError("expected alias after AS")
            return A.ExprWithAlias(expr, A.Ident(v, '"' if k == "qident" else None))
        k, v = self.peek()
        if k == "word" and v.upper() not in _STOP_WORDS:
            self.next()
            return A.ExprWithAlias(expr, A.Ident(v))
        return A.UnnamedExpr(expr)

    def parse_select(self) -> Select:
        if not self.accept_word("SELECT"):
            raise SqlParseError("expected SELECT")
        items = [self.parse_select_item()]
        while self.accept_op(","):
            items.append(self.parse_select_item())
        from_ = None
        if self.accept_word("FROM"):
            k, name = self.next()
            if k != "word":
                raise SqlParseError("expected table function or table name after FROM")
            args: List[str] = []
            if self.accept_op("("):
                while not self.accept_op(")"):
                    kk, vv = self.next()
                    if kk == "eof":
                        raise SqlParseError("unterminated table function")
                    if kk == "string":
                        args.append(vv)
            alias = None
            if self.accept_word("AS"):
                alias = self.next()[1]
            else:
                k2, v2 = self.peek()
                if k2 == "word" and v2.upper() not in _STOP_WORDS:
                    alias = self.next()[1]
            from_ = TableFunc(name, tuple(args), alias)
        selection = None
        if self.accept_word("WHERE"):
            selection = self.parse_expr(0)
        self.accept_op(";")
        return Select(tuple(items), from_, selection)


def parse_expr(text: str) -> A.Expr:
    """Parse one SQL scalar expression (what follows WHERE, or one select item's expression)."""
    p = _Parser(_tokenize(text))
    e = p.parse_expr(0)
    if p.peek()[0] != "eof":
        raise SqlParseError(f"trailing tokens after expression: {p.peek()[1]!r}")
    return e


def parse_select(text: str) -> Select:
    p = _Parser(_tokenize(text))
    s = p.parse_select()
    if p.peek()[0] != "eof":
        raise SqlParseError(f"trailing tokens after statement: {p.peek()[1]!r}")
    return s


def parse_statements(text: str) -> List[Select]:
    """Split on ';' like the reference's multi-statement handling (planner/test_sqlparser_behavior.rs)."""
    toks = _tokenize(text)
    stmts, cur = [], []
    for t in toks:
        if t == ("op", ";"):
            if cur:
                stmts.append(cur)
            cur = []
        else:
            cur.append(t)
    if cur:
        stmts.append(cur)
    return [_Parser(s).parse_select() for s in stmts]
