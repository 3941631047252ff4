"""String side of the formula path: what happens to generated token ids after the decoder.

`UniMERNetDecode.token2str` (rapid_doc/model/formula/rapid_formula_self/model_handler/pp_formulanet_plus/
post_process.py:277-296) cuts every sequence at the first EOS (id 2, inclusive), BPE-decodes it, then runs `post_process`
(:350-381): `remove_chinese_text_wrapping`, `fix_latex` = `fix_latex_left_right(fix_delimiter=False)` ->
`fix_latex_environments` -> `remove_up_commands` -> `remove_unsupported_commands` (.../utils.py:9-49,253-312), and
`ftfy.fix_text`.  The tokenizer JSON is download-only and ftfy is a third-party package absent here, so those two steps
stay with the caller (`FormulaRecognizer(token_decoder=...)`); everything between them is restated below and pinned by
tests/golden/latex_post_seed*.json, minted by calling the reference functions on seeded LaTeX token soups (including their
quirks: `\\leftarrow` opens a `\\left`, `align*` is used as an unescaped regex, a moved `\\right.` is re-inserted at the
index computed BEFORE its removal).
"""
from __future__ import annotations

import re
from typing import List, Sequence

import numpy as np

EOS_ID = 2


def cut_at_eos(token_ids: Sequence[Sequence[int]]) -> List[np.ndarray]:
    """post_process.py:286-291: keep everything up to and including the first EOS."""
    out = []
    for row in token_ids:
        row = np.asarray(row)
        hit = np.nonzero(row == EOS_ID)[0]
        out.append(row[: int(hit[0]) + 1] if len(hit) else row)
    return out


# ---------------------------------------------------------------------------------------------------------------------
def _odd_backslashes_before(s: str, i: int) -> bool:
    n = 0
    j = i - 1
    while j >= 0 and s[j] == "\\":
        n += 1
        j -= 1
    return n % 2 == 1


def _group_end(s: str, start: int, depth: int) -> int:
    """Index of the `}` that closes the brace group of nesting `depth` open at `start` (utils.py:134-148), or -1."""
    level = depth
    for i in range(start, len(s)):
        ch = s[i]
        if ch == "{" and not _odd_backslashes_before(s, i):
            level += 1
        elif ch == "}" and not _odd_backslashes_before(s, i):
            level -= 1
            if level < depth:
                return i
    return -1


def _regroup_left_right(s: str) -> str:
    """utils.py:51-131: a `\\right?` whose `\\left?` was opened at another brace depth is moved to the end of that group."""
    n = len(s)
    depth = 0                 # only the NUMBER of open braces matters to the reference's stack
    opened = []               # (position of \left, brace depth there)
    moves = []                # (start, end, target) of a \right? to relocate
    i = 0
    while i < n:
        if i > 0 and s[i - 1] == "\\" and _odd_backslashes_before(s, i):
            i += 1                                   # escaped character
            continue
        if i + 5 < n and s.startswith("\\left", i):
            opened.append((i, depth))
            i += 6                                   # the command and its delimiter character
            continue
        if i + 6 < n and s.startswith("\\right", i):
            if opened:
                pos, d = opened.pop()
                if d != depth:
                    target = _group_end(s, pos, d)
                    if target != -1:
                        moves.append((i, i + 7, target))
            i += 7
            continue
        if s[i] == "{":
            depth += 1
        elif s[i] == "}" and depth > 0:
            depth -= 1
        i += 1
    if not moves:
        return s
    cells = list(s)                                   # one character per cell; an inserted piece occupies ONE cell,
    for start, end, target in sorted(moves, key=lambda m: m[0], reverse=True):   # exactly like the reference's list
        piece = "".join(cells[start:end])
        del cells[start:end]
        cells.insert(target, piece)
    return "".join(cells)


_LEFT_CMD = re.compile(r"\\left(?![a-zA-Z])")
_RIGHT_CMD = re.compile(r"\\right(?![a-zA-Z])")
_LEFT_RIGHT_STRIP = re.compile(r"\\left\.?|\\right\.?")


def fix_left_right(s: str) -> str:
    """`fix_latex_left_right(s, fix_delimiter=False)` (utils.py:9-49)."""
    if len(_LEFT_CMD.findall(s)) == len(_RIGHT_CMD.findall(s)):
        return _regroup_left_right(s)
    return _LEFT_RIGHT_STRIP.sub("", s)


_ENVS = ("array", "matrix", "pmatrix", "bmatrix", "vmatrix", "Bmatrix", "Vmatrix", "cases", "aligned", "gathered", "align",
         "align*")
# the environment name goes into the pattern unescaped, as in the reference (utils.py:247-251): `align*` is a regex there
_ENV_RE = {e: (re.compile(r"\\begin\{" + e + r"\}"), re.compile(r"\\end\{" + e + r"\}"),
               re.compile(r"\\begin\{" + e + r"\}\{([^}]*)\}")) for e in _ENVS}


def fix_environments(s: str) -> str:
    """utils.py:253-278: prepend missing `\\begin{env}` (with the column format of the first one found), append missing `\\end`."""
    for env in _ENVS:
        begin_re, end_re, fmt_re = _ENV_RE[env]
        nb, ne = len(begin_re.findall(s)), len(end_re.findall(s))
        if ne > nb:
            m = fmt_re.search(s)
            fmt = "{" + m.group(1) + "}" if m else ("{c}" if env == "array" else "")
            s = ("\\begin{" + env + "}" + fmt + " ") * (ne - nb) + s
        elif nb > ne:
            s = s + (" \\end{" + env + "}") * (nb - ne)
    return s


_UP = re.compile(r"\\up([a-zA-Z]+)")
_KEEP_UP = ("arrow", "downarrow", "lus", "silon")
_DROP = re.compile(r"\\(?:lefteqn|boldmath|ensuremath|centering|textsubscript|sides|textsl|textcent|emph|protect|null)")
_CJK_TEXT = re.compile(r"\\text\s*{\s*([^}]*?[\u4e00-\u9fff]+[^}]*?)\s*}")


def strip_up_prefix(s: str) -> str:
    """utils.py:298-304: `\\upalpha` -> `\\alpha`; `\\uparrow`, `\\updownarrow`, `\\uplus`, `\\upsilon` stay."""
    return _UP.sub(lambda m: m.group(0) if m.group(1) in _KEEP_UP else "\\" + m.group(1), s)


def drop_unsupported(s: str) -> str:
    """utils.py:307-312."""
    return _DROP.sub("", s)


def unwrap_cjk_text(s: str) -> str:
    """post_process.py:340-348: `\\text{...CJK...}` loses its wrapper; double quotes are removed."""
    return _CJK_TEXT.sub(lambda m: m.group(1), s).replace('"', "")


def latex_postprocess(text: str) -> str:
    """`UniMERNetDecode.post_process` without its final `ftfy.fix_text` (post_process.py:350-381)."""
    return drop_unsupported(strip_up_prefix(fix_environments(fix_left_right(unwrap_cjk_text(text)))))
