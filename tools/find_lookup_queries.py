#!/usr/bin/env python3
"""Recovers which advice / fixed queries the ATMS-with-lookups circuit's lookup argument reads, from the reference's
gates_test.hbs vectors alone (tests/golden/reference_kats.json): exhaustive search over every assignment of
(tag, sel, val, t_tag, t_val) against lookup_expression_3_1.  Prints the matching assignments (exactly one).
Used once to write tests/test_oracle_golden.py: ATMS_LOOKUP_QUERIES."""
import json, itertools, sys
R=0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
k=json.load(open('tests/golden/reference_kats.json'))['lookup_identities']
i={n:int(v,16) for n,v in k['inputs'].items()}
e={n:int(v,16) for n,v in k['expected'].items()}
adv=[i['advice_eval_%d'%j] for j in range(1,12)]
fix=[i['fixed_eval_%d'%j] for j in range(1,22)]
th,be,ga=i['theta'],i['beta'],i['gamma']
prod,pn,pin,pinv,ptab=i['product_eval_1'],i['product_next_eval_1'],i['permuted_input_eval_1'],i['permuted_input_inv_eval_1'],i['permuted_table_eval_1']
act=i['active_rows']
left=pn*(pin+be)%R*(ptab+ga)%R
want=e['lookup_expression_3_1']
# want = (left - prod*(inp+be)*(tab+ga))*act  =>  prod*(inp+be)*(tab+ga) = left - want/act
target=(left - want*pow(act,R-2,R))%R
target=target*pow(prod,R-2,R)%R   # (inp+be)*(tab+ga)
# tab candidates
tabs={}
for a in range(21):
    for b in range(21):
        tabs[(fix[a]*th+fix[b]+ga)%R]=(a,b)
found=[]
for tg in range(21):
    for sl in range(21):
        for v in range(11):
            inp=(fix[tg]*th+fix[sl]*adv[v])%R
            need=target*pow((inp+be)%R,R-2,R)%R
            if need in tabs:
                found.append((tg,sl,v,tabs[need]))
print(found)
