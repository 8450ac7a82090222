"""Minimal stand-in for the ``pytest_check`` plugin (not installed here), covering what the reference's tests/check_ot_result.py
calls: soft assertions become hard ones."""


def equal(a, b, msg=""):
    assert a == b, msg


def is_true(x, msg=""):
    assert x, msg


def is_false(x, msg=""):
    assert not x, msg


def is_none(x, msg=""):
    assert x is None, msg


def is_not_none(x, msg=""):
    assert x is not None, msg
