"""Answer-string metrics used by ``SDNetTrainer.predict`` (restated from Utils/eval_func.py:1-35, 62-68).
Host string work on the decoded answers; not on the accelerated path."""


def stvqa_score(a, b):
    """1 - Levenshtein(a, b) / max(len): the ANLS per-pair score."""
    a, b = a.lower(), b.lower()
    if max(len(a), len(b)) == 0:
        return 1
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return 1 - prev[-1] / max(len(a), len(b))


def note_stvqa(gt_list, word):
    return max([stvqa_score(gt, word) for gt in gt_list] + [-1])


def note_textvqa(gt_list, word):
    return sum(1 for gt in gt_list if gt.lower() == word) / 10.0
