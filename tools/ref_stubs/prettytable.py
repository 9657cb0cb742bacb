class PrettyTable:
    def __init__(self): self.field_names=[]; self.rows=[]
    def add_row(self, r): self.rows.append(r)
    def __str__(self): return str(self.field_names)+"\n"+"\n".join(str(r) for r in self.rows)
