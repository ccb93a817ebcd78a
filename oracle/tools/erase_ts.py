#!/usr/bin/env python3
"""Fixture-generation tool (TEST INFRASTRUCTURE, container-only).

Mechanically erases TypeScript type syntax from the reference's 15 hot-path
`.ts` files so that they parse under the Node 12 present in the build container
(there is no tsc/esbuild here).  No semantic edits: only type syntax is removed,
the four `??` uses are rewritten to an equivalent ternary, ESM import/export is
rewritten to CommonJS.  This file contains syntax rules only - no reference code.

Its OUTPUT is derived reference code: it is written under /tmp (BBQ_ERASED_OUT),
never under /root/repo, never committed, never shipped.  Only the golden
vectors that oracle/tools/gen_fixtures.js produces by RUNNING that output are
committed (tests/golden/).  Recipe: SURVEY.md Appendix B.
"""
import re, sys, os

SRC = os.environ.get("BBQ_REF_SRC", "/root/reference/src")
OUT = os.environ.get("BBQ_ERASED_OUT", "/tmp/bbq_ref_erased/js")
FILES = ['types.ts', 'constants.ts', 'utils.ts', 'vectorOperations.ts', 'vectorSimilarity.ts',
         'vectorUtils.ts', 'minHeap.ts', 'utils/bitcount.ts',
         'utils/computeBatchFourBitDotProductDirectPacked.ts', 'bitwiseDotProduct.ts',
         'batchDotProduct.ts', 'optimizedScalarQuantizer.ts', 'binaryQuantizedScorer.ts',
         'binaryQuantizationFormat.ts', 'topKSelector.ts']


def skip_ws(s, i):
    while i < len(s) and s[i] in ' \t\r\n':
        i += 1
    return i


def skip_balanced(s, i, open_c, close_c):
    assert s[i] == open_c, (s[i - 20:i + 20])
    depth = 0
    while i < len(s):
        c = s[i]
        if c in '\'"`':
            q = c
            i += 1
            while s[i] != q:
                if s[i] == '\\':
                    i += 1
                i += 1
        elif c == open_c:
            depth += 1
        elif c == close_c:
            depth -= 1
            if depth == 0:
                return i + 1
        i += 1
    raise ValueError('unbalanced')


def skip_type(s, i):
    """s[i:] starts a type expression; return index just past it."""
    i = skip_ws(s, i)
    while True:
        # primary
        c = s[i]
        if c == '{':
            i = skip_balanced(s, i, '{', '}')
        elif c == '[':
            i = skip_balanced(s, i, '[', ']')
        elif c == '(':
            i = skip_balanced(s, i, '(', ')')
            j = skip_ws(s, i)
            if s.startswith('=>', j):
                i = skip_type(s, j + 2)
        elif c in '\'"':
            i = s.index(c, i + 1) + 1
        else:
            m = re.compile(r'[A-Za-z_$][\w$]*(\s*\.\s*[A-Za-z_$][\w$]*)*').match(s, i)
            if not m:
                raise ValueError('type? ' + repr(s[i - 30:i + 30]))
            word = m.group(0)
            i = m.end()
            if word in ('typeof', 'keyof', 'readonly'):
                continue
            if i < len(s) and s[i] == '<':
                i = skip_balanced(s, i, '<', '>')
        # array suffixes
        while s.startswith('[]', i):
            i += 2
        j = skip_ws(s, i)
        if j < len(s) and s[j] in '|&' and not s.startswith('||', j) and not s.startswith('&&', j):
            i = skip_ws(s, j + 1)
            continue
        # type predicate:  x is T
        if s.startswith('is ', j):
            i = skip_ws(s, j + 3)
            continue
        return i


def erase_param_list(s, i):
    """s[i] == '(' of a declaration parameter list. Returns (new_text, end_index)."""
    end = skip_balanced(s, i, '(', ')')
    body = s[i + 1:end - 1]
    out = []
    k = 0
    while k < len(body):
        k0 = k
        k = skip_ws(body, k)
        out.append(body[k0:k])
        if k >= len(body):
            break
        # strip modifiers
        m = re.compile(r'((public|private|protected|readonly)\s+)*').match(body, k)
        k = m.end()
        m = re.compile(r'(\.\.\.)?[A-Za-z_$][\w$]*').match(body, k)
        if not m:
            # destructuring etc. not used in the reference's declarations
            raise ValueError('param? ' + repr(body[k:k + 40]))
        out.append(m.group(0))
        k = m.end()
        k2 = skip_ws(body, k)
        if k2 < len(body) and body[k2] == '?':
            k2 = skip_ws(body, k2 + 1)
        if k2 < len(body) and body[k2] == ':':
            k = skip_type(body + ' ', k2 + 1)
            k2 = skip_ws(body, k)
        if k2 < len(body) and body[k2] == '=':
            # default value: copy through to next top-level comma
            depth = 0
            j = k2
            while j < len(body):
                ch = body[j]
                if ch in '([{':
                    depth += 1
                elif ch in ')]}':
                    depth -= 1
                elif ch == ',' and depth == 0:
                    break
                j += 1
            default = body[k2:j]
            # typed arrow inside default (minHeap.ts:17)
            default = re.sub(r'\((\w+): T, (\w+): T\)', r'(\1, \2)', default)
            out.append(' ' + default)
            k = j
        else:
            k = k2
        if k < len(body) and body[k] == ',':
            out.append(',')
            k += 1
    return '(' + ''.join(out) + ')', end


DECL = re.compile(
    r'(?m)^(?P<indent>[ \t]*)(?P<head>(?:export\s+)?(?:(?:public|private|protected)\s+)?(?:static\s+)?(?:async\s+)?'
    r'(?:function\s+[A-Za-z_$][\w$]*|constructor|(?!if\b|for\b|while\b|switch\b|return\b|catch\b|throw\b|else\b|new\b|function\b)[A-Za-z_$][\w$]*))'
    r'(?P<generic><[A-Za-z, ]+>)?\s*\(')


def erase_declarations(s):
    out = []
    pos = 0
    for m in DECL.finditer(s):
        if m.start() < pos:
            continue
        head = m.group('head')
        paren = m.end() - 1
        try:
            end = skip_balanced(s, paren, '(', ')')
        except ValueError:
            continue
        j = skip_ws(s, end)
        is_decl = False
        ret_end = end
        if j < len(s) and s[j] == '{':
            is_decl = True
        elif j < len(s) and s[j] == ':':
            try:
                t = skip_type(s, j + 1)
                t2 = skip_ws(s, t)
                if t2 < len(s) and s[t2] == '{':
                    is_decl = True
                    ret_end = t
            except ValueError:
                pass
        if not is_decl:
            continue
        # a call statement like `foo(...) {`? not in this codebase. Exclude keywords handled by regex.
        if not (head.startswith(('export', 'public', 'private', 'protected', 'static', 'function', 'constructor', 'async'))
                or re.match(r'^[A-Za-z_$][\w$]*$', head)):
            continue
        params, _ = erase_param_list(s, paren)
        head2 = re.sub(r'\b(public|private|protected)\s+', '', head)
        out.append(s[pos:m.start()])
        out.append(m.group('indent') + head2 + params)
        pos = ret_end
    out.append(s[pos:])
    return ''.join(out)


def erase_var_annotations(s):
    pat = re.compile(r'(?m)^([ \t]*(?:export\s+)?(?:const|let|var)\s+[A-Za-z_$][\w$]*)\s*:\s*')
    out = []
    pos = 0
    for m in pat.finditer(s):
        if m.start() < pos:
            continue
        t = skip_type(s, m.end())
        out.append(s[pos:m.start()])
        out.append(m.group(1))
        pos = t
    out.append(s[pos:])
    s = ''.join(out)
    # for (let i: number = ...) not used. Destructured catch etc fine.
    return s


def erase_class_fields(s):
    lines = s.split('\n')
    res = []
    for ln in lines:
        m = re.match(r'^(\s*)((?:public|private|protected)\s+)?(static\s+)?(readonly\s+)?([A-Za-z_$][\w$]*)\s*:\s*(.*)$', ln)
        if m and (m.group(2) or m.group(4)):
            rest = m.group(6)
            # find end of type
            try:
                t = skip_type(rest + ' ;', 0)
            except ValueError:
                res.append(ln)
                continue
            tail = rest[t:].strip()
            if tail.startswith('='):
                res.append(f"{m.group(1)}{m.group(3) or ''}{m.group(5)} {tail}")
            else:
                res.append(m.group(1) + '// [type-erased field] ' + m.group(5))
            continue
        m = re.match(r'^(\s*)(?:public|private|protected)\s+(static\s+)([A-Za-z_$][\w$]*)\s*=\s*(.*)$', ln)
        if m:
            res.append(f"{m.group(1)}{m.group(2)}{m.group(3)} = {m.group(4)}")
            continue
        res.append(ln)
    return '\n'.join(res)


def remove_blocks(s, kw):
    pat = re.compile(r'(?m)^[ \t]*(?:export\s+)?' + kw + r'\s+[A-Za-z_$][\w$]*[^{]*\{')
    while True:
        m = pat.search(s)
        if not m:
            return s
        end = skip_balanced(s, m.end() - 1, '{', '}')
        s = s[:m.start()] + s[end:]


def strip_comments_keep_lines(s):
    # blank out comments so colons/keywords in prose never confuse the eraser; keeps line numbers
    def repl(m):
        t = m.group(0)
        if t.startswith('/'):
            return re.sub(r'[^\n]', ' ', t)
        return t
    pat = re.compile(r'//[^\n]*|/\*.*?\*/|\'(?:\\.|[^\'\\\n])*\'|"(?:\\.|[^"\\\n])*"|`(?:\\.|[^`\\])*`', re.S)
    return pat.sub(repl, s)


def convert(rel):
    s = open(os.path.join(SRC, rel), encoding='utf-8').read()
    s = strip_comments_keep_lines(s)
    exports = []
    # --- imports
    s = re.sub(r'(?s)import\s+type\s*\{[^}]*\}\s*from\s*\'[^\']+\';?', '', s)
    s = re.sub(r'(?s)import\s*\{([^}]*)\}\s*from\s*\'([^\']+)\';?',
               lambda m: 'const {' + m.group(1) + '} = require(\'' + m.group(2) + '\');', s)
    s = re.sub(r'export\s*\{(\w+)\}\s*from\s*\'([^\']+)\'',
               lambda m: (exports.append(m.group(1)) or '') + 'const {' + m.group(1) + '} = require(\'' + m.group(2) + '\');', s)
    # --- interfaces / enums
    s = remove_blocks(s, 'interface')

    def enum_repl(m):
        name = m.group(1)
        body = m.group(2)
        exports.append(name)
        items = re.findall(r'(\w+)\s*=\s*(\'[^\']*\')', body)
        return 'const ' + name + ' = {' + ', '.join(f'{k}: {v}' for k, v in items) + '};'
    s = re.sub(r'(?s)export\s+enum\s+(\w+)\s*\{(.*?)\}', enum_repl, s)
    # --- class heads
    s = re.sub(r'class\s+(\w+)<T>', r'class \1', s)
    s = re.sub(r'\s+implements\s+\w+', '', s)
    # --- generics on constructor calls / static types
    s = re.sub(r'new\s+(MinHeap|WeakMap|Map|Array)<', lambda m: 'new ' + m.group(1) + '\x00<', s)
    while '\x00<' in s:
        i = s.index('\x00<')
        end = skip_balanced(s, i + 1, '<', '>')
        s = s[:i] + s[end:]
    # --- `as X`
    s = re.sub(r'\}\s*as\s+const', '}', s)
    s = re.sub(r'\(a as any\) - \(b as any\)', '(a) - (b)', s)
    s = re.sub(r'\s+as\s+any\b', '', s)
    # --- declarations (params + return types), var annotations, fields
    s = erase_class_fields(s)
    s = erase_declarations(s)
    s = erase_var_annotations(s)
    # arrow with type predicate (topKSelector.ts:110)
    s = s.replace('(candidate): candidate is TopKCandidate =>', '(candidate) =>')
    # --- non-null assertions  x!  x[i]!  f()!
    s = re.sub(r'(?<=[\w\)\]])!(?=[\s\.\)\],;\[\+\-\*/]|$)(?!=)', '', s, flags=re.M)
    # --- `a ?? b`  (operands in the reference are side-effect-free member reads)
    s = re.sub(r'= (config\.\w+) \?\? ([\w\.]+);', r'= (\1 !== undefined && \1 !== null) ? \1 : \2;', s)
    s = s.replace('centroid[i] = vectors[0][i] ?? 0;',
                  'centroid[i] = (vectors[0][i] !== undefined && vectors[0][i] !== null) ? vectors[0][i] : 0;')
    # --- exports
    s = re.sub(r'(?m)^export\s+((?:function|class|const)\s+([A-Za-z_$][\w$]*))', lambda m: (exports.append(m.group(2)) or m.group(1)), s)
    s += '\n' + ''.join(f'exports.{e} = {e};\n' for e in exports)
    dst = os.path.join(OUT, rel[:-3] + '.js')
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    open(dst, 'w', encoding='utf-8').write("'use strict';\n" + s)
    return dst


if __name__ == '__main__':
    for f in FILES:
        try:
            print('ok ', convert(f))
        except Exception as e:
            print('ERR', f, e)
