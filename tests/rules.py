"""Hand-derived expectations for the arrow-rs 53 semantics the reference relies on but does not pin with
its own tests ("unpinned-by-reference").  Every expected value below was
worked out from the documented behaviour, NOT produced by the oracle or the GPU library; both are tested
against this table (tests/test_oracle_golden.py on CPU, tests/test_gpu_parity.py on the GPU).
"""
from __future__ import annotations

import math

import numpy as np
import pyarrow as pa

NAN = float("nan")
INF = float("inf")


def _f32(vals, mask=None):
    return pa.array(np.array(vals, dtype=np.float32), mask=None if mask is None else np.array(mask))


def _neg_nan32():
    return np.frombuffer(np.uint32(0xFFC00000).tobytes(), dtype=np.float32)[0]


def batch_floats():
    x = np.array([-0.0, 0.0, NAN, INF, -INF, 1.5], dtype=np.float32)
    x = np.concatenate([x, np.array([_neg_nan32()], dtype=np.float32)])
    y = np.array([0.0, -0.0, NAN, INF, 0.0, 1.5, NAN], dtype=np.float32)
    return pa.RecordBatch.from_arrays([pa.array(x), pa.array(y), pa.array(x.astype(np.float64))], names=["x", "y", "d"])


def batch_ints():
    return pa.RecordBatch.from_arrays([
        pa.array([1, -7, 2147483647, -2147483648, 0, 9], pa.int32()),
        pa.array([3, 2, 1, -1, 5, 0], pa.int32()),
        pa.array([100, 200, 255, 0, 1, 2], pa.uint8()),
        pa.array([1, 2, 3, 4, 5, 6], pa.int64()),
        pa.array([1, 2, 3, 4, 5, 6], pa.uint32()),
        pa.array([-128, 127, 5, -5, 0, 1], pa.int8()),
    ], names=["i", "j", "u8", "l", "u32", "i8"])


def batch_nulls():
    return pa.RecordBatch.from_arrays([
        pa.array([1, None, 3, 4, None, 6], pa.int32()),
        pa.array([10, 20, None, 40, None, 0], pa.int32()),
        pa.array([True, None, False, True, None, False], pa.bool_()),
        pa.array(["a", None, "c", "dd", "e", None], pa.utf8()),
        pa.array([1.5, 2.5, None, None, 5.5, 6.5], pa.float32()),
    ], names=["a", "b", "t", "s", "f"])


def batch_min():
    """MIN of every signed width next to a -1 and a 7 of the same type (a literal -1 cannot be written: the reference has
    no unary minus, compute_value.rs:338-342)."""
    cols, names = [], []
    for name, typ, lo in (("m8", pa.int8(), -2**7), ("m16", pa.int16(), -2**15), ("m32", pa.int32(), -2**31), ("m64", pa.int64(), -2**63)):
        cols += [pa.array([7, lo, 9], typ), pa.array([-1, -1, -1], typ), pa.array([7, 13, 9], typ)]
        names += [name, name.replace("m", "n"), name.replace("m", "p")]
    return pa.RecordBatch.from_arrays(cols, names=names)


def batch_strings():
    return pa.RecordBatch.from_arrays([
        pa.array(["b", "a", "", "ab", "abc", "bé", "B"], pa.utf8()),
        pa.array(["b", "b", "b", "aa", "ab", "b", "b"], pa.utf8()),
        pa.array([0, 1, 2, 3, 4, 5, 6], pa.int32()),
    ], names=["s", "r", "id"])


def batch_temporal():
    """Same-type temporal / decimal pairs (compared on their raw i32 / i64 / i128 values) next to columns whose DataType
    differs in unit, time zone, precision or scale only."""
    D = __import__("decimal").Decimal
    big = 2**70
    return pa.RecordBatch.from_arrays([
        pa.array([0, 19000, -5, None, 7, 7], pa.date32()), pa.array([1, 18999, -5, 3, None, 6], pa.date32()),
        pa.array([0, 86400000, -86400000, 5, 5, 6], pa.date64()), pa.array([0, 0, 0, 5, 6, 5], pa.date64()),
        pa.array([10, 20, 2**40, -2**40, None, 0], pa.timestamp("s")), pa.array([10, 21, 2**40 - 1, -2**40 + 1, 3, 0], pa.timestamp("s")),
        pa.array([10, 20, 30, 40, 50, 60], pa.timestamp("ms")),
        pa.array([10, 20, 30, 40, 50, 60], pa.timestamp("s", tz="UTC")),
        pa.array([1, 2, 3, 4, 5, 6], pa.time32("s")), pa.array([6, 5, 3, 2, 1, 0], pa.time32("s")),
        pa.array([1, 2, 3, 4, 5, 6], pa.time64("us")), pa.array([6, 5, 3, 2, 1, 0], pa.time64("us")),
        pa.array([-1, 0, 1, 2, 3, None], pa.duration("ms")), pa.array([0, 0, 0, 3, 3, 3], pa.duration("ms")),
        pa.array([D(big) / 100, D(-big) / 100, D("1.50"), D("-0.01"), None, D(big + 1) / 100], pa.decimal128(30, 2)),
        pa.array([D(big + 1) / 100, D(-big - 1) / 100, D("1.50"), D("0.00"), D("1.00"), D(big) / 100], pa.decimal128(30, 2)),
        pa.array([D("1.500")] * 6, pa.decimal128(30, 3)),
        pa.array([1, 2, 3, 4, 5, 6], pa.int32()),
    ], names=["d1", "d2", "e1", "e2", "ts1", "ts2", "tms", "tz", "t1", "t2", "u1", "u2", "du1", "du2", "dec1", "dec2", "dec3", "id"])


def _f16(vals):
    return pa.array(np.array(vals, dtype=np.float16), pa.float16())


def _f16_bits(bits):
    return pa.array(np.array(bits, dtype=np.uint16).view(np.float16), pa.float16())


def batch_halves():
    return pa.RecordBatch.from_arrays([
        #          -0.0    0.0     NaN     -NaN    inf     1.5     2048    65504 (max)  6e-8 (min subnormal)  0.1 (rounded)
        _f16_bits([0x8000, 0x0000, 0x7E00, 0xFE00, 0x7C00, 0x3E00, 0x6800, 0x7BFF, 0x0001, 0x2E66]),
        _f16_bits([0x0000, 0x8000, 0x7E00, 0x7E00, 0x7C00, 0x3E00, 0x3C00, 0x7BFF, 0x0001, 0x3C00]),
        pa.array(np.array([0.0, 0.0, 1.0, 1.0, 1e38, 1.5, 2048.5, 65504.0, 6e-8, 0.1], dtype=np.float32)),
        pa.array(np.array([0.0, 0.0, 1.0, 1.0, 1e300, 1.5, 2048.5, 65504.0, 6e-8, 0.1], dtype=np.float64)),
        pa.array([1, 2, 3, 4, 5, 6, 7, 8, 9, 10], pa.int32()),
        pa.array([True, True, True, False, True, None, True, True, True, True], pa.bool_()),
    ], names=["h", "k", "f", "d", "i", "t"])


def batch_bool_words():
    return pa.RecordBatch.from_arrays([
        pa.array(["true", " YES ", "0", "off", "maybe", None, "\u2003t\u00a0", "TRU", "", "of", "tr ue", "1", "\tfAlSe\n", "no\u3000", "2", "yess"], pa.utf8()),
        pa.array([True, True, True, True, True, True, False, None, True, True, True, False, True, True, True, True], pa.bool_()),
        pa.array(list(range(16)), pa.int32()),
    ], names=["s", "b", "id"])


def batch_one_row():
    return pa.RecordBatch.from_arrays([pa.array([7], pa.uint64()), pa.array(["yes"], pa.utf8())], names=["b", "w"])


# (name, batch factory, kind, sql / select, expectation)
# kind "value": expectation = (arrow type, python list)   [compute_value]
# kind "filter": expectation = list of surviving row indices of the input batch
# kind "error": expectation = status code
# kind "project": expectation = dict name -> (type, values) in output order, plus "nullable" dict
RULES = [
    # ---- IEEE totalOrder comparisons (arrow-ord cmp; ArrowNativeTypeOp::compare = total_cmp) -----------------
    ("float_lt_zero_total_order", batch_floats, "value", "x < 0.0", (pa.bool_(), [True, False, False, False, True, False, True])),
    ("float_eq_is_bitwise", batch_floats, "value", "x = y", (pa.bool_(), [False, False, True, True, False, True, False])),
    ("float_neq_is_bitwise", batch_floats, "value", "x <> y", (pa.bool_(), [True, True, False, False, True, False, True])),
    ("float_gt_nan_is_greatest", batch_floats, "value", "x > 1000000.0", (pa.bool_(), [False, False, True, True, False, False, False])),
    ("float_lteq_total_order", batch_floats, "value", "x <= y", (pa.bool_(), [True, False, True, True, True, True, True])),
    ("float_gteq_total_order", batch_floats, "value", "x >= y", (pa.bool_(), [False, True, True, True, False, True, False])),
    ("float64_total_order", batch_floats, "value", "d < 0.0", (pa.bool_(), [True, False, False, False, True, False, True])),
    # ---- float arithmetic is plain IEEE --------------------------------------------------------------------
    ("float_div_by_zero", batch_floats, "value", "x / 0.0", (pa.float32(), [NAN, NAN, NAN, INF, -INF, INF, NAN])),
    ("float_rem_is_fmod", batch_ints, "value", "i % 2.5", (pa.float32(), [1.0, -2.0, math.fmod(2147483648.0, 2.5), -math.fmod(2147483648.0, 2.5), 0.0, 1.5])),
    # ---- checked integer arithmetic ------------------------------------------------------------------------
    ("int_add_overflow_errors", batch_ints, "error", "i + 1", 20),
    ("int_mul_overflow_errors", batch_ints, "error", "i * 2", 20),
    ("int_div_by_zero_errors", batch_ints, "error", "u8 / (u8 % 2)", 21),
    ("int64_div_by_zero_errors", batch_ints, "error", "l / (l % 2)", 21),
    ("int_rem_by_zero_errors", batch_ints, "error", "j % (j % 1)", 21),
    ("first_offending_row_decides_the_error", batch_ints, "error", "i / j", 20),
    ("int_div_truncates", batch_ints, "value", "i / 2", (pa.int32(), [0, -3, 1073741823, -1073741824, 0, 4])),
    ("int_rem_sign_of_dividend", batch_ints, "value", "i % 4", (pa.int32(), [1, -3, 3, 0, 0, 1])),
    # MIN / -1 and MIN % -1 (unpinned-by-reference; arrow-rs is not in the container).  Encoded behaviour = arrow-arith 53
    # numeric.rs: `Op::Div => try_op!(.., l.div_checked(r))`, `Op::Rem => try_op!(.., l.mod_checked(r))`, and
    # ArrowNativeTypeOp::{div,mod}_checked for integers = zero check (DivideByZero) then `checked_div` / `checked_rem`,
    # both None for MIN op -1 => ArrowError::ArithmeticOverflow("Overflow happened on: MIN % -1"); the doc comment of
    # numeric::rem says "Overflow or division by zero will result in an error".  The alternative reading (the pre-numeric
    # arithmetic::modulus kernel: zero check + mod_wrapping => MIN % -1 = 0) is what the round-1 review recalled; if a
    # maintainer shows arrow 53.x does that, flip these four `rem` rows to a "value" row with 0 and change
    # kernels.hip `Interp::arith` / chq_oracle.c INT_ARITH together.
    ("i8_min_div_neg1_overflows", batch_min, "error", "m8 / n8", 20),
    ("i16_min_div_neg1_overflows", batch_min, "error", "m16 / n16", 20),
    ("i32_min_div_neg1_overflows", batch_min, "error", "m32 / n32", 20),
    ("i64_min_div_neg1_overflows", batch_min, "error", "m64 / n64", 20),
    ("i8_min_rem_neg1_overflows", batch_min, "error", "m8 % n8", 20),
    ("i16_min_rem_neg1_overflows", batch_min, "error", "m16 % n16", 20),
    ("i32_min_rem_neg1_overflows", batch_min, "error", "m32 % n32", 20),
    ("i64_min_rem_neg1_overflows", batch_min, "error", "m64 % n64", 20),
    ("rem_neg1_is_zero_away_from_min", batch_min, "value", "p32 % n32", (pa.int32(), [0, 0, 0])),
    ("div_neg1_negates_away_from_min", batch_min, "value", "p64 / n64", (pa.int64(), [-7, -13, -9])),
    ("i8_rem_neg1_is_zero_away_from_min", batch_min, "value", "p8 % n8", (pa.int8(), [0, 0, 0])),
    ("uint8_add_checked", batch_ints, "error", "u8 + u8", 20),
    ("int8_mul_checked", batch_ints, "error", "i8 * i8", 20),
    ("widening_u8_to_i32", batch_ints, "value", "u8 + 1000", (pa.int32(), [1100, 1200, 1255, 1000, 1001, 1002])),
    ("widening_i32_to_i64", batch_ints, "value", "j * l", (pa.int64(), [3, 4, 3, -4, 25, 0])),
    ("i32_big_literal_is_i64", batch_ints, "value", "j + 3000000000", (pa.int64(), [3000000003, 3000000002, 3000000001, 2999999999, 3000000005, 3000000000])),
    ("int_to_f32_coercion", batch_ints, "value", "j / 2.0", (pa.float32(), [1.5, 1.0, 0.5, -0.5, 2.5, 0.0])),
    ("constant_subexpression_stays_scalar", batch_ints, "filter", "i > 25 + 0.0", [2]),
    # ---- coercion table misses -----------------------------------------------------------------------------
    ("i64_with_f32_unsupported", batch_ints, "error", "l + 1.5", 9),
    ("u32_with_i32_unsupported", batch_ints, "error", "u32 + 1", 9),
    ("bool_with_int_unsupported", batch_nulls, "error", "t = 1", 9),
    ("utf8_with_int_unsupported", batch_strings, "error", "s = 1", 9),
    ("utf8_arithmetic_invalid", batch_strings, "error", "s + r", 22),
    ("minus_not_implemented", batch_ints, "error", "i - 1", 3),
    ("unary_minus_not_implemented", batch_ints, "error", "i > -5", 2),
    ("column_not_found", batch_ints, "error", "nope > 1", 7),
    ("compound_identifier_not_found", batch_ints, "error", "t.i > 1", 8),
    ("three_part_identifier_not_found", batch_ints, "error", "a.b.c > 1", 8),
    ("long_number_not_implemented", batch_ints, "error", "i > 5L", 1),
    ("null_literal_not_implemented", batch_ints, "error", "i = null", 1),
    ("filter_needs_boolean", batch_ints, "error_filter", "i + 1000000000000", 10),
    # ---- nulls: non-Kleene and/or, union of validities, null mask slot = dropped ----------------------------
    ("null_propagates_through_add", batch_nulls, "value", "a + b", (pa.int32(), [11, None, None, 44, None, 6])),
    ("null_propagates_through_cmp", batch_nulls, "value", "a < b", (pa.bool_(), [True, None, None, True, None, False])),
    ("and_is_not_kleene", batch_nulls, "value", "t and a > 100", (pa.bool_(), [False, None, False, False, None, False])),
    ("or_is_not_kleene", batch_nulls, "value", "t or a > 0", (pa.bool_(), [True, None, True, True, None, True])),
    ("null_mask_rows_are_dropped", batch_nulls, "filter", "a < b", [0, 3]),
    ("null_div_zero_slot_is_not_an_error", batch_nulls, "value", "b / a", (pa.int32(), [10, None, None, 10, None, 0])),
    ("int_and_casts_to_bool", batch_nulls, "value", "a and b", (pa.bool_(), [True, None, None, True, None, False])),
    ("float_and_casts_to_bool", batch_nulls, "filter", "f and t", [0]),
    ("utf8_eq_with_nulls", batch_nulls, "value", "s = 'c'", (pa.bool_(), [False, None, True, False, False, None])),
    ("filter_keeps_nulls_in_other_columns", batch_nulls, "filter", "f > 2.0", [1, 4, 5]),
    # ---- Utf8 comparisons are byte-lexicographic -------------------------------------------------------------
    ("utf8_lt_scalar", batch_strings, "value", "s < 'b'", (pa.bool_(), [False, True, True, True, True, False, True])),
    ("utf8_gteq_scalar", batch_strings, "value", "s >= 'ab'", (pa.bool_(), [True, False, False, True, True, True, False])),
    ("utf8_scalar_on_the_left", batch_strings, "value", "'ab' < s", (pa.bool_(), [True, False, False, False, True, True, False])),
    ("utf8_column_vs_column", batch_strings, "value", "s <= r", (pa.bool_(), [True, True, True, False, False, False, True])),
    ("utf8_filter_mixed", batch_strings, "filter", "s <> 'b' and id % 2 = 0", [2, 4, 6]),
    # ---- same-type temporal / decimal comparisons (get_common_type's `left == right` arm, compute_value.rs:355; arrow-ord
    # compares the native i32 / i64 / i128 values).  unpinned-by-reference ------------------------------------------
    ("date32_lt", batch_temporal, "value", "d1 < d2", (pa.bool_(), [True, False, False, None, None, False])),
    ("date32_gteq", batch_temporal, "value", "d1 >= d2", (pa.bool_(), [False, True, True, None, None, True])),
    ("date64_eq", batch_temporal, "value", "e1 = e2", (pa.bool_(), [True, False, False, True, False, False])),
    ("date64_lt_negative", batch_temporal, "value", "e1 < e2", (pa.bool_(), [False, False, True, False, True, False])),
    ("timestamp_lteq", batch_temporal, "value", "ts1 <= ts2", (pa.bool_(), [True, True, False, True, None, True])),
    ("timestamp_neq", batch_temporal, "value", "ts1 <> ts2", (pa.bool_(), [False, True, True, True, None, False])),
    ("time32_gt", batch_temporal, "value", "t1 > t2", (pa.bool_(), [False, False, False, True, True, True])),
    ("time64_lteq", batch_temporal, "value", "u1 <= u2", (pa.bool_(), [True, True, True, False, False, False])),
    ("duration_lt", batch_temporal, "value", "du1 < du2", (pa.bool_(), [True, False, False, True, False, None])),
    ("decimal128_lt_uses_all_128_bits", batch_temporal, "value", "dec1 < dec2", (pa.bool_(), [True, False, False, True, None, False])),
    ("decimal128_eq", batch_temporal, "value", "dec1 = dec2", (pa.bool_(), [False, False, True, False, None, False])),
    ("decimal128_gteq", batch_temporal, "value", "dec2 >= dec1", (pa.bool_(), [True, False, True, True, None, False])),
    ("date32_filter_keeps_every_column", batch_temporal, "filter", "d1 < d2", [0]),
    ("decimal128_filter", batch_temporal, "filter", "dec1 < dec2 or id = 3", [0, 2, 3]),
    ("timestamp_filter_mixed", batch_temporal, "filter", "ts1 <= ts2 and id > 1", [1, 3, 5]),
    ("timestamp_units_differ", batch_temporal, "error", "ts1 < tms", 9),
    ("timestamp_time_zones_differ", batch_temporal, "error", "ts1 < tz", 9),
    ("decimal_scales_differ", batch_temporal, "error", "dec1 < dec3", 9),
    ("date32_vs_date64", batch_temporal, "error", "d1 < e1", 9),
    ("date_vs_literal", batch_temporal, "error", "d1 < 5", 9),
    ("date_plus_date_invalid", batch_temporal, "error", "d1 + d2", 22),
    ("time_times_time_invalid", batch_temporal, "error", "t1 * t2", 22),
    ("duration_rem_invalid", batch_temporal, "error", "du1 % du2", 22),
    ("date_under_and_cannot_cast", batch_temporal, "error", "d1 and id > 1", 24),
    # ---- Float16: total order on the raw halves, exact widening, arithmetic = f32 operation rounded back to f16 -----
    ("f16_lt_total_order", batch_halves, "value", "h < k", (pa.bool_(), [True, False, False, True, False, False, False, False, False, True])),
    ("f16_eq_is_bitwise", batch_halves, "value", "h = k", (pa.bool_(), [False, False, True, False, True, True, False, True, True, False])),
    ("f16_gteq_total_order", batch_halves, "value", "h >= k", (pa.bool_(), [False, True, True, False, True, True, True, True, True, False])),
    ("f16_widens_to_f32", batch_halves, "value", "h < f", (pa.bool_(), [True, False, False, True, False, False, True, False, True, True])),
    ("f16_widens_to_f64", batch_halves, "value", "h = d", (pa.bool_(), [False, True, False, False, False, True, False, True, False, False])),
    ("f16_with_f32_literal", batch_halves, "value", "k * 0.5", (pa.float32(), [0.0, -0.0, NAN, NAN, INF, 0.75, 0.5, 32752.0, 2.9802322387695312e-08, 0.5])),
    ("f16_add_rounds_to_nearest_even", batch_halves, "value", "h + k", (pa.float16(), [0.0, 0.0, NAN, NAN, INF, 3.0, 2048.0, INF, 1.1920928955078125e-07, 1.0996094])),
    ("f16_mul", batch_halves, "value", "h * k", (pa.float16(), [-0.0, -0.0, NAN, NAN, INF, 2.25, 2048.0, INF, 0.0, 0.099975586])),
    ("f16_div_by_zero_is_ieee", batch_halves, "value", "k / h", (pa.float16(), [NAN, NAN, NAN, NAN, NAN, 1.0, 0.00048828125, 1.0, 1.0, 10.0])),
    ("f16_filter", batch_halves, "filter", "h < k", [0, 3, 9]),
    ("f16_under_and_casts_to_bool", batch_halves, "value", "h and t", (pa.bool_(), [False, False, True, False, True, None, True, True, True, True])),
    ("f16_with_int_unsupported", batch_halves, "error", "h + i", 9),
    ("f16_with_int_literal_unsupported", batch_halves, "error", "h < 1", 9),
    # ---- Utf8 under AND / OR: arrow-cast's string -> Boolean (compute_value.rs:72-73, :95-96), invalid spelling = NULL ---
    ("utf8_and_casts_spellings", batch_bool_words, "value", "s and b",
     (pa.bool_(), [True, True, False, False, None, None, False, None, None, False, None, False, False, False, None, None])),
    ("utf8_or_casts_spellings", batch_bool_words, "value", "s or id > 100",
     (pa.bool_(), [True, True, False, False, None, None, True, True, None, False, None, True, False, False, None, None])),
    ("utf8_predicate_filter", batch_bool_words, "filter", "s and id > 0", [1, 6, 7, 11]),
    ("utf8_literals_under_and", batch_bool_words, "value", "'yes' and 'off'", (pa.bool_(), [False])),
    ("utf8_bad_literal_is_null", batch_bool_words, "value", "'maybe' or 'true'", (pa.bool_(), [None])),
    ("utf8_literal_vs_column_length_mismatch", batch_bool_words, "error", "'true' and id > 0", 23),
    # a one-row batch is the only place where a literal-built array meets a column without a length error
    ("utf8_null_literal_next_to_a_one_row_column", batch_one_row, "value", "b and ('b' and 2.5)", (pa.bool_(), [None])),
    ("utf8_literal_next_to_a_one_row_column", batch_one_row, "value", "w and ('on' or false)", (pa.bool_(), [True])),
    ("utf8_null_literal_filters_nothing", batch_one_row, "filter", "b and ('b' and 2.5)", []),
    # ---- and / or take plain BooleanArrays: no scalar broadcast, result is never a scalar ---------------------
    ("and_with_literal_length_mismatch", batch_ints, "error", "i > 0 and true", 23),
    ("or_with_literal_length_mismatch", batch_ints, "error", "false or i > 0", 23),
    ("and_of_literals_is_len1_array", batch_ints, "value", "true and false", (pa.bool_(), [False])),
    ("len1_array_vs_column_length_mismatch", batch_ints, "error", "(true and true) = (i > 0)", 22),
    # ---- scalar predicate quirk: a literal-only mask has length 1, arrow filters the first row only ------------
    ("where_true_keeps_first_row_only", batch_ints, "filter", "true", [0]),
    ("where_false_keeps_nothing", batch_ints, "filter", "false", []),
    ("where_constant_comparison", batch_ints, "filter", "1 < 2", [0]),
]

def batch_minus():
    return pa.RecordBatch.from_arrays([
        pa.array([5, 0, 200], pa.uint8()), pa.array([3, 1, 100], pa.uint8()), pa.array([7, 2, 201], pa.uint8()),
        pa.array([-128, 127, 5], pa.int8()), pa.array([1, -1, 5], pa.int8()),
        pa.array([-2**63, 5, 2**63 - 1], pa.int64()), pa.array([1, 7, -1], pa.int64()),
        pa.array([0, 5, 2**64 - 1], pa.uint64()), pa.array([1, 5, 0], pa.uint64()),
        pa.array([1.5, 0.25, -4.0], pa.float64()),
    ], names=["a", "b", "c", "x8", "y8", "x64", "y64", "ux", "uy", "d"])


# BinaryOperator::Minus -- NOT reference behaviour (compute_value.rs:210-216 has no Minus arm; "minus_not_implemented"
# above is the reference's answer).  These rows describe the product's opt-in `enable_minus` option = arrow-arith
# numeric::sub: checked for integers, IEEE for floats, same coercion / null / scalar rules as Plus.  Checked against the
# oracle's clearly labelled extension mode (O.extension_minus) on the CPU and against the library on the GPU.
MINUS_RULES = [
    ("minus_i32", batch_ints, "value", "j - 1", (pa.int32(), [2, 1, 0, -2, 4, -1])),
    ("minus_i32_overflow", batch_ints, "error", "i - 1", 20),
    ("minus_literal_on_the_left", batch_ints, "value", "10 - j", (pa.int32(), [7, 8, 9, 11, 5, 10])),
    ("minus_float_literal_on_the_left", batch_ints, "value", "2.5 - j", (pa.float32(), [-0.5, 0.5, 1.5, 3.5, -2.5, 2.5])),
    ("minus_u8_widens_with_literal", batch_ints, "value", "u8 - 1", (pa.int32(), [99, 199, 254, -1, 0, 1])),
    ("minus_u8_same_type", batch_minus, "value", "c - a", (pa.uint8(), [2, 2, 1])),
    ("minus_u8_underflow", batch_minus, "error", "a - b", 20),
    ("minus_i8_overflow", batch_minus, "error", "x8 - y8", 20),
    ("minus_i8_to_i32_no_overflow", batch_minus, "value", "x8 - 1", (pa.int32(), [-129, 126, 4])),
    ("minus_i64_overflow_low", batch_minus, "error", "x64 - y64", 20),
    ("minus_i64", batch_minus, "value", "y64 - 3000000000", (pa.int64(), [-2999999999, -2999999993, -3000000001])),
    ("minus_u64_underflow", batch_minus, "error", "ux - uy", 20),
    ("minus_u64", batch_minus, "value", "ux - ux", (pa.uint64(), [0, 0, 0])),
    ("minus_f64", batch_minus, "value", "d - 0.5", (pa.float64(), [1.0, -0.25, -4.5])),
    ("minus_f32_ieee", batch_floats, "value", "x - y", (pa.float32(), [-0.0, 0.0, NAN, NAN, -INF, 0.0, NAN])),
    ("minus_nulls_union", batch_nulls, "value", "a - b", (pa.int32(), [-9, None, None, -36, None, 6])),
    ("minus_constants_fold_to_scalar", batch_ints, "value", "5 - 3", (pa.int32(), [2])),
    ("minus_constant_overflow", batch_ints, "error", "0 - 2147483647 - 2", 20),
    ("minus_in_predicate", batch_ints, "filter", "i > 10 - 1", [2]),
    ("minus_mixed_with_other_ops", batch_ints, "value", "j * 2 - j / 1", (pa.int32(), [3, 2, 1, -1, 5, 0])),
]

PROJECT_RULES = [
    # name, batch factory, select list sql, expected [(name, type, values, nullable)]
    ("naming_unnamed_counts_identifiers", batch_ints,
     "select i, j + 1, i8, j * 2, l as big from t",
     [("i", pa.int32(), [1, -7, 2147483647, -2147483648, 0, 9], False),
      ("unnamed_1", pa.int32(), [4, 3, 2, 0, 6, 1], False),
      ("i8", pa.int8(), [-128, 127, 5, -5, 0, 1], False),
      ("unnamed_3", pa.int32(), [6, 4, 2, -2, 10, 0], False),
      ("big", pa.int64(), [1, 2, 3, 4, 5, 6], False)]),
    ("nullable_follows_null_count", batch_nulls,
     "select a + 1 as a1, f, t and t as tt, s from t",
     [("a1", pa.int32(), [2, None, 4, 5, None, 7], True),
      ("f", pa.float32(), [1.5, 2.5, None, None, 5.5, 6.5], True),
      ("tt", pa.bool_(), [True, None, False, True, None, False], True),
      ("s", pa.utf8(), ["a", None, "c", "dd", "e", None], True)]),
    ("wildcard_then_expr", batch_strings,
     "select *, id * id as sq from t",
     [("s", pa.utf8(), ["b", "a", "", "ab", "abc", "bé", "B"], True),
      ("r", pa.utf8(), ["b", "b", "b", "aa", "ab", "b", "b"], True),
      ("id", pa.int32(), [0, 1, 2, 3, 4, 5, 6], True),
      ("sq", pa.int32(), [0, 1, 4, 9, 16, 25, 36], False)]),
]

PROJECT_ERRORS = [
    ("qualified_wildcard_not_implemented", batch_ints, "select t.* from t", 11),
    ("scalar_projection_length_mismatch", batch_ints, "select i, 1 + 2 from t", 22),
    ("projection_error_propagates", batch_ints, "select i * i from t", 20),
]
