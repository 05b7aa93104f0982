cd /root/repo
python __graft_entry__.py smoke 2>&1 | tail -2
mv oracle/_ref oracle/_ref_hidden; python __graft_entry__.py smoke 2>&1 | tail -2; mv oracle/_ref_hidden oracle/_ref
