#!/usr/bin/env python3
"""Extract the Kalbar observation sheets of the reference's xlsx data files into CSV fixtures
(tests/golden/data/), with the Python standard library only: this image has no openpyxl, and
the reference's own loader (Data_Import.py:457-458,:529-530,:590,:645) also fails on pandas
2.x (`sheetname=`).  Run in the build container, where /root/reference exists:

    python tests/golden/make_locinfo_fixtures.py

Copies the two plain-text Kalbar geometry files next to them.  Data only -- no reference code.
"""
import csv
import datetime
import os
import re
import shutil
import sys
import xml.etree.ElementTree as ET
import zipfile

REF = os.environ.get('PARASITOIDS_REF', '/root/reference')
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data')
NS = {'m': 'http://schemas.openxmlformats.org/spreadsheetml/2006/main',
      'r': 'http://schemas.openxmlformats.org/officeDocument/2006/relationships',
      'p': 'http://schemas.openxmlformats.org/package/2006/relationships'}
SHEETS = [('sampling_details.xlsx', 'Kal-sentinels-raw', 'kalbar_sentinels_raw.csv'),
          ('sampling_details.xlsx', 'Kal-releasefield-raw', 'kalbar_releasefield_raw.csv'),
          ('adult_counts_kalbar.xlsx', 'adult counts field A', 'kalbar_adult_counts_field_A.csv'),
          ('adult_counts_kalbar.xlsx', 'cardinal 15 mar 05', 'kalbar_cardinal_15mar05.csv'),
          ('adult_counts_kalbar.xlsx', 'cardinal 21 mar 05', 'kalbar_cardinal_21mar05.csv')]
TEXT = ['kalbarfields.txt', 'kalbarreleasegrid.txt']
BUILTIN_DATE = set(range(14, 23)) | {45, 46, 47}


def col_index(ref):
    n = 0
    for ch in re.match(r'[A-Z]+', ref).group(0):
        n = n * 26 + ord(ch) - 64
    return n - 1


def read_sheet(path, sheet_name):
    z = zipfile.ZipFile(path)
    wb = ET.fromstring(z.read('xl/workbook.xml'))
    rels = ET.fromstring(z.read('xl/_rels/workbook.xml.rels'))
    target = {r.get('Id'): r.get('Target') for r in rels.findall('p:Relationship', NS)}
    rid = None
    for s in wb.find('m:sheets', NS):
        if s.get('name') == sheet_name:
            rid = s.get('{%s}id' % NS['r'])
    if rid is None:
        raise KeyError(sheet_name)
    strings = []
    if 'xl/sharedStrings.xml' in z.namelist():
        for si in ET.fromstring(z.read('xl/sharedStrings.xml')).findall('m:si', NS):
            strings.append(''.join(t.text or '' for t in si.iter('{%s}t' % NS['m'])))
    styles = ET.fromstring(z.read('xl/styles.xml'))
    custom = {int(n.get('numFmtId')): n.get('formatCode') for n in styles.iter('{%s}numFmt' % NS['m'])}
    is_date = []
    for xf in styles.find('m:cellXfs', NS):
        fid = int(xf.get('numFmtId', '0'))
        code = re.sub(r'"[^"]*"|\[[^\]]*\]', '', custom.get(fid, ''))
        is_date.append(fid in BUILTIN_DATE or bool(re.search(r'[dmy]', code, re.I)) and fid in custom)
    rows = []
    sheet = ET.fromstring(z.read('xl/' + target[rid].lstrip('/').replace('xl/', '')))
    for row in sheet.find('m:sheetData', NS):
        vals = {}
        for c in row:
            v = c.find('m:v', NS)
            if v is None or v.text is None:
                inline = c.find('m:is', NS)
                if inline is None:
                    continue
                text = ''.join(t.text or '' for t in inline.iter('{%s}t' % NS['m']))
            elif c.get('t') == 's':
                text = strings[int(v.text)]
            elif c.get('t') in ('str', 'b', 'e'):
                text = v.text
            else:
                text = v.text
                if c.get('s') is not None and is_date[int(c.get('s'))]:
                    day = float(v.text)
                    d = datetime.datetime(1899, 12, 30) + datetime.timedelta(days=day)
                    text = d.strftime('%Y-%m-%d') if day == int(day) else d.isoformat()
                else:
                    f = float(v.text)
                    text = str(int(f)) if f == int(f) else repr(f)
            vals[col_index(c.get('r'))] = text
        if vals:
            rows.append(vals)
    width = max(max(r) for r in rows) + 1
    return [[r.get(i, '') for i in range(width)] for r in rows]


def main():
    os.makedirs(OUT, exist_ok=True)
    for book, sheet, out in SHEETS:
        rows = read_sheet(os.path.join(REF, 'data', book), sheet)
        # drop columns that are empty in the header row and in every data row
        keep = [i for i in range(len(rows[0])) if any(r[i] != '' for r in rows)]
        with open(os.path.join(OUT, out), 'w', newline='') as f:
            w = csv.writer(f)
            for r in rows:
                w.writerow([r[i] for i in keep])
        print('%-34s %4d rows x %2d cols  header: %s' % (out, len(rows) - 1, len(keep), [rows[0][i] for i in keep]))
    for t in TEXT:
        shutil.copyfile(os.path.join(REF, 'data', t), os.path.join(OUT, t))
        print('copied', t)


if __name__ == '__main__':
    sys.exit(main())
