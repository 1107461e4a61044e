"""The usage snippet in README.md runs as written."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_readme_python_snippet_runs(fiksi):
    text = open(os.path.join(ROOT, "README.md")).read()
    code = re.search(r"```python\n(.*?)```", text, re.S).group(1)
    scope = {}
    exec(compile(code, "README.md", "exec"), scope)
    assert len(scope["results"]) == 100_000 and (scope["results"]["sse_unscaled"] < 1e-4).mean() > 0.98
