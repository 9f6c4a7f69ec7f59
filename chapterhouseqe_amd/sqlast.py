"""Host-side mirror of the `sqlparser::ast` subset that ChapterhouseDB's record_utils consume.

The reference hands `sqlparser 0.52` AST nodes (`Expr`, `SelectItem`) to
`compute_value` / `filter_record` / `project_record`
(reference: src/handlers/operator_handler/operators/record_utils/compute_value.rs:57-61,
filter_record.rs:21-25, record_projection.rs:16-20).  These classes carry the same variants and
field names so tests and callers read like the reference's own (`Expr::BinaryOp { left, op, right }`
-> `BinaryOp(left, op, right)`).  Variants compute_value does not implement are kept as
`Unsupported*` nodes so the error path (`ExpressionTypeNotImplemented`, ...) is reproducible.
"""
from __future__ import annotations

import enum
from dataclasses import dataclass
from typing import Optional, Tuple, Union


class BinaryOperator(enum.Enum):
    """sqlparser::ast::BinaryOperator (subset + the ones that must hit the not-implemented arm)."""
    Plus = "Plus"
    Minus = "Minus"
    Multiply = "Multiply"
    Divide = "Divide"
    Modulo = "Modulo"
    StringConcat = "StringConcat"
    Gt = "Gt"
    Lt = "Lt"
    GtEq = "GtEq"
    LtEq = "LtEq"
    Spaceship = "Spaceship"
    Eq = "Eq"
    NotEq = "NotEq"
    And = "And"
    Or = "Or"
    Xor = "Xor"
    BitwiseOr = "BitwiseOr"
    BitwiseAnd = "BitwiseAnd"
    BitwiseXor = "BitwiseXor"


@dataclass(frozen=True)
class Ident:
    value: str
    quote_style: Optional[str] = None


# ---- sqlparser::ast::Value -------------------------------------------------------------------
@dataclass(frozen=True)
class Number:
    """Value::Number(String, bool) -- text kept verbatim; typing happens in compute_value.rs:219-250."""
    text: str
    long: bool = False


@dataclass(frozen=True)
class SingleQuotedString:
    value: str


@dataclass(frozen=True)
class Boolean:
    value: bool


@dataclass(frozen=True)
class UnsupportedValue:
    """Any other Value variant (Null, DoubleQuotedString, HexStringLiteral, ...)."""
    debug: str


Value = Union[Number, SingleQuotedString, Boolean, UnsupportedValue]


# ---- sqlparser::ast::Expr --------------------------------------------------------------------
class Expr:
    pass


@dataclass(frozen=True)
class Identifier(Expr):
    ident: Ident


@dataclass(frozen=True)
class CompoundIdentifier(Expr):
    idents: Tuple[Ident, ...]


@dataclass(frozen=True)
class ValueExpr(Expr):
    """Expr::Value(v)"""
    value: Value


@dataclass(frozen=True)
class BinaryOp(Expr):
    left: Expr
    op: BinaryOperator
    right: Expr


@dataclass(frozen=True)
class Nested(Expr):
    expr: Expr


@dataclass(frozen=True)
class UnsupportedExpr(Expr):
    """UnaryOp, Function, Case, ... -- compute_value.rs:338-342 rejects them."""
    debug: str


# ---- sqlparser::ast::SelectItem --------------------------------------------------------------
class SelectItem:
    pass


@dataclass(frozen=True)
class Wildcard(SelectItem):
    pass


@dataclass(frozen=True)
class QualifiedWildcard(SelectItem):
    prefix: str


@dataclass(frozen=True)
class UnnamedExpr(SelectItem):
    expr: Expr


@dataclass(frozen=True)
class ExprWithAlias(SelectItem):
    expr: Expr
    alias: Ident


# ---- small constructors used by tests (the reference builds these structs literally) ----------
def ident(name: str) -> Identifier:
    return Identifier(Ident(name))


def compound(*parts: str) -> CompoundIdentifier:
    return CompoundIdentifier(tuple(Ident(p) for p in parts))


def number(text: str, long: bool = False) -> ValueExpr:
    return ValueExpr(Number(text, long))


def string(value: str) -> ValueExpr:
    return ValueExpr(SingleQuotedString(value))


def boolean(value: bool) -> ValueExpr:
    return ValueExpr(Boolean(value))


def binop(left: Expr, op: BinaryOperator, right: Expr) -> BinaryOp:
    return BinaryOp(left, op, right)
