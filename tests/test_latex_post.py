"""String fix-ups of the formula path against vectors minted by the reference's own functions
(tests/golden/make_golden.py: latex_post_golden) - 1200 LaTeX token soups, strings must be identical."""
import json

import numpy as np
import pytest

from rapiddoc_amd import latex_post as L


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_latex_postprocess_matches_reference(golden_dir, seed):
    g = json.loads((golden_dir / f"latex_post_seed{seed}.json").read_text())
    assert len(g["inputs"]) == 400
    moved = 0
    for text, ref in zip(g["inputs"], g["outputs"]):
        assert L.latex_postprocess(text) == ref
        t = L.unwrap_cjk_text(text)
        moved += len(L._LEFT_CMD.findall(t)) == len(L._RIGHT_CMD.findall(t)) and L._regroup_left_right(t) != t
    assert moved >= 10        # the corpus really exercises the \right relocation


def test_known_answers():
    # unequal \left / \right counts: every \left, \right (and a following dot) is stripped; \leftarrow is not counted
    assert L.fix_left_right(r"\left( a \right) \right.") == "( a ) "
    assert L.fix_left_right(r"\leftarrow \left( a \right)") == r"\leftarrow \left( a \right)"
    # \right in a deeper group than its \left: moved to the end of the group that holds the \left (utils.py:51-131)
    assert L.fix_left_right(r"{\left( a {b \right) c} d} e") != r"{\left( a {b \right) c} d} e"
    assert L.fix_environments(r"a \end{array}") == r"\begin{array}{c} a \end{array}"
    assert L.fix_environments(r"\begin{cases} a") == r"\begin{cases} a \end{cases}"
    assert L.strip_up_prefix(r"\upalpha \uparrow \uplus \upsilon \updownarrow") == r"\alpha \uparrow \uplus \upsilon \updownarrow"
    assert L.drop_unsupported(r"\emph{a}\protect\null b") == "{a} b"
    assert L.unwrap_cjk_text('\\text{中文} "x"') == "中文 x"


def test_cut_at_eos():
    out = L.cut_at_eos(np.array([[0, 5, 6, 2, 1, 1], [0, 7, 8, 9, 10, 11], [2, 3, 2, 1, 1, 1]]))
    assert [o.tolist() for o in out] == [[0, 5, 6, 2], [0, 7, 8, 9, 10, 11], [2]]
